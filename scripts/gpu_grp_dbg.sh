#!/bin/bash
# timing experiments on sparse_inv_group_kernel (VR_SPARSE_GROUP_DBG bits: 1 no adds, 2 no emit, 4 prologue only, 8 loads only)
set -o pipefail
R=/root/repo; O=$R/gpurun_out/grp; mkdir -p $O
cd $R
for m in 0 2 1 9 4; do
  VR_SPARSE_GROUP=${GRP:-4} VR_SPARSE_GROUP_DBG=$m timeout -k 10 200 python scripts/perf_hybrid_batch.py 1000000 1000 3 sparse_only > $O/dbg_$m.txt 2>&1 || exit 1
  echo "== dbg $m"; grep -A1 'sparse batch' $O/dbg_$m.txt
done
