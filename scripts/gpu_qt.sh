#!/bin/bash
# A/B of vr_query_text's forms on bench.py's own from-text section (profiles/r03_experiments.md §13):
# lane + mask before the forward pass (default) | the forward pass, then vr_search_hybrid (VR_QUERY_TEXT_LANE_FIRST=0) |
# the sparse leg queued before the forward pass as well (VR_QUERY_TEXT_AHEAD=1)
set -o pipefail
R=/root/repo; O=$R/gpurun_out/qt; mkdir -p $O
cd $R
for v in "1 0" "0 0" "1 1" "1 0" "0 0" "1 1"; do
  set -- $v
  VR_QUERY_TEXT_LANE_FIRST=$1 VR_QUERY_TEXT_AHEAD=$2 VR_BENCH_TEXT_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --aniso-rows 0 --other-rows 0 > $O/bench_$1$2.json 2> $O/bench_$1$2.err || exit 1
  echo "lane_first=$1 ahead=$2"; grep 'text debug' $O/bench_$1$2.err
done
