#!/bin/bash
set -o pipefail
R=/root/repo; O=$R/gpurun_out/grp; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_batch_hybrid_gpu.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/perf_hybrid_batch.py > $O/perf_p4.txt 2>&1 || exit 1
sed -n 6,10p $O/perf_p4.txt
