#!/bin/bash
# A/B of vr_query_text with and without the sparse leg queued before the forward pass (VR_QUERY_TEXT_AHEAD), on bench.py's own from-text section (profiles/r03_experiments.md §13)
set -o pipefail
R=/root/repo; O=$R/gpurun_out/qt; mkdir -p $O
cd $R
for a in 0 1 0 1; do
  VR_QUERY_TEXT_AHEAD=$a VR_BENCH_TEXT_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --aniso-rows 0 --other-rows 0 > $O/bench_$a.json 2> $O/bench_$a.err || exit 1
  echo "ahead=$a"; grep 'text debug' $O/bench_$a.err
done
