import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, 'qps', d.get('qps_batched_1k'), 'ms/call', d.get('ms_per_batched_call'), 'recall', d.get('recall_at_10_batched_vs_torch_matmul'))
    print('   stats', d['roofline_search'].get('two_stage'))
    print('   aniso', d.get('anisotropic_corpus'))
    print('   batched roofline', d.get('roofline_batched_search',{}).get('avg_launch_ms'), d.get('roofline_batched_search',{}).get('share_of_call_time'))
