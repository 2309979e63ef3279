#!/bin/bash
# The grouped sparse batch scan (csrc/invert.hip) on the GPU box: its parity tests (incl. the 1.25M-row configs[4] share), the
# timings of scripts/perf_hybrid_batch.py, the seed-pass variant, and the time split with parts of the scan switched off
# (VR_SPARSE_GROUP_DBG; profiles/r03_experiments.md §12). Outputs under gpurun_out/grp/.
set -o pipefail
R=/root/repo; O=$R/gpurun_out/grp; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_batch_hybrid_gpu.py tests/test_fullsize_gpu.py -x -q -k "batch or hybrid or sparse" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for g in 4; do
  VR_SPARSE_GROUP=$g timeout -k 10 200 python scripts/perf_hybrid_batch.py > $O/perf_g$g.txt 2>&1 || exit 1
  echo "== group $g"; sed -n 6,10p $O/perf_g$g.txt
done
VR_SPARSE_GROUP_SAMPLE=0 timeout -k 10 200 python scripts/perf_hybrid_batch.py 1000000 1000 3 sparse_only > $O/nosample.txt 2>&1 || exit 1
echo "== seed instead of sample"; grep -A2 'sparse batch' $O/nosample.txt
for m in 0 2 1 9 4; do
  VR_SPARSE_GROUP_DBG=$m timeout -k 10 200 python scripts/perf_hybrid_batch.py 1000000 1000 3 sparse_only > $O/dbg_$m.txt 2>&1 || exit 1
  echo "== dbg $m"; grep -A2 'sparse batch' $O/dbg_$m.txt
done
