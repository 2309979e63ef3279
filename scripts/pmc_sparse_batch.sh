#!/bin/bash
# Hardware-counter passes over the grouped sparse batch scan (separate rocprofv3 --pmc runs, few counters each; 1M rows x 40
# stems, 1000 queries of 4-6 Zipf terms: scripts/perf_hybrid_batch.py ... sparse_only). Averages per launch of
# sparse_inv_group_kernel<4, false>, counters summed over XCDs / SEs. Output: gpurun_out/pmc_sparse/summary.txt
R=/root/repo; O=$R/gpurun_out/pmc_sparse; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS" \
           ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python $R/scripts/perf_hybrid_batch.py 1000000 1000 1 sparse_only > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
done
python - <<'PY' | tee /root/repo/gpurun_out/pmc_sparse/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/root/repo/gpurun_out/pmc_sparse/p*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        key = (r['Dispatch_Id'], r['Counter_Name'])
        per[key] += float(r['Counter_Value']); names[r['Dispatch_Id']] = r['Kernel_Name']
    for (d, c), v in per.items():
        n = names[d]
        k = None
        if 'sparse_inv_group_kernel' in n: k = 'sparse_inv_group_kernel<%s>' % n.split('<')[1].split('>')[0]
        elif 'sparse_inv_locate' in n: k = 'sparse_inv_locate_kernel'
        elif 'select_regions' in n: k = 'select_regions_kernel'
        if k:
            acc[k][c][0] += 1; acc[k][c][1] += v
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n, v = acc[k][c]
        print(f"   {c:36s} {v / n:16.1f}  (n={n})")
PY
rm -rf $O/p*/
