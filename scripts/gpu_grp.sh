#!/bin/bash
# grouped sparse batch: parity tests, then the perf script per group size, then one rocprofv3 kernel-stats pass
set -o pipefail
R=/root/repo; O=$R/gpurun_out/grp; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_batch_hybrid_gpu.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for g in ${GROUPS_TO_RUN:-8 4 2}; do
  VR_SPARSE_GROUP=$g timeout -k 10 200 python scripts/perf_hybrid_batch.py > $O/perf_g$g.txt 2>&1 || exit 1
  echo "== group $g"; sed -n 6,10p $O/perf_g$g.txt
done
cd /tmp && export TMPDIR=/tmp
VR_SPARSE_GROUP=${PROF_GROUP:-4} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/perf_hybrid_batch.py 1000000 1000 3 > $O/prof.log 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob
for f in glob.glob('/root/repo/gpurun_out/grp/prof/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if any(x in n for x in ('sparse_inv', 'select_counted', 'expand', 'batch_weights')) and 'sparse_inv_kernel' not in n:
            print(n[:48], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
