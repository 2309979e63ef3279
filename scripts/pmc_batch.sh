#!/bin/bash
# Hardware-counter passes over the batched dense search (separate rocprofv3 --pmc runs, few counters each).
R=/root/repo; O=$R/gpurun_out/pmc_batch; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" \
           ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python $R/scripts/perf_batch.py 200000 1000 1 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('/root/repo/gpurun_out/pmc_batch/p*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        key = (r['Dispatch_Id'], r['Counter_Name'])
        per[key] += float(r['Counter_Value']); names[r['Dispatch_Id']] = r['Kernel_Name']
    for (d, c), v in per.items():
        n = names[d]
        if 'batch_scan_kernel' in n:
            k = 'batch_scan<%s>' % n.split('<')[1].split('>')[0]
            acc[k][c][0] += 1; acc[k][c][1] += v
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n, v = acc[k][c]
        print(f"   {c:36s} {v / n:16.1f}  (n={n})")
PY
rm -rf $O/p*/
