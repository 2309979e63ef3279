#!/bin/bash
# occupancy experiment on the grouped sparse scan: posting slots per thread (4 vs 8: registers) x queries per group
# (build the twin first, here: make -C voitta_rag_amd/csrc slices8; profiles/r03_experiments.md §15)
set -o pipefail
R=/root/repo; O=$R/gpurun_out/grp; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_batch_hybrid_gpu.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for lib in libvoitta_engine.so libvoitta_engine_s8.so; do
  for g in 2 4; do
    VOITTA_ENGINE_LIB=$R/voitta_rag_amd/$lib VR_SPARSE_GROUP=$g timeout -k 10 200 python scripts/perf_hybrid_batch.py 1000000 1000 3 sparse_only > $O/occ_${lib}_$g.txt 2>&1 || exit 1
    echo "== $lib group $g"; grep -A1 'sparse batch' $O/occ_${lib}_$g.txt
  done
done
