#!/bin/bash
set -o pipefail
R=/root/repo; O=$R/gpurun_out/grp; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_batch_hybrid_gpu.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
VR_SPARSE_GROUP_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dropin-files 0 --aniso-rows 0 --other-rows 0 > $O/bench_dbg.json 2> $O/bench_dbg.err; echo "bench rc=$?"
grep 'sparse grouped' $O/bench_dbg.err | sort | uniq -c | head -20
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/grp/bench_dbg.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ["qps_hybrid_batched_1k", "ms_per_hybrid_batched_call", "recall_at_10_hybrid_batched_vs_single_query"]}, d['roofline_search']['two_stage'])
PY
