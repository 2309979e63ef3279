#!/bin/bash
# One GPU-box call while iterating: the tests named in $1 (a -k expression, or "all"), then the hybrid-batch timing table,
# then its per-kernel statistics. Outputs under gpurun_out/step/.
set -o pipefail
R=/root/repo; O=$R/gpurun_out/step; mkdir -p $O; rm -f $O/*
cd $R
if [ "$1" = "all" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$1" > $O/tests.log 2>&1
fi
echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 300 python scripts/perf_hybrid_batch.py > $O/perf_hybrid.log 2>&1; echo "perf rc=$?"; cat $O/perf_hybrid.log | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/scripts/perf_hybrid_batch.py 1000000 1000 3 > $O/stats.log 2>&1; echo "rocprof rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -25 $O/kernel_stats.csv | cut -c1-150
find $O/stats -name "*kernel_trace.csv" -delete
cd $R && VR_BATCH_STAMPS=1 timeout -k 10 200 python scripts/perf_batch.py 1000000 1000 2 2>&1 | grep -v amdgpu.ids | tail -6
cd $R && VR_SPARSE_DEBUG=1 timeout -k 10 200 python scripts/perf_hybrid_batch.py 1000000 1000 1 2>&1 | grep "sparse batch\]" | sort | uniq -c | head -5
