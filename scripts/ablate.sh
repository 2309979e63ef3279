#!/bin/bash
# timing-only ablations of gemm_f16x3_kernel (VR_GEMM_ABLATE: 1 = no stores, 2 = no steady-state loads, 4 = no MFMA)
export PYTHONPATH=.
for a in 0 1 2 4 3 6 7; do
  echo -n "ablate=$a  "
  VR_GEMM_ABLATE=$a timeout -k 10 200 python scripts/perf_encode.py bge-base-en-v1.5 256 128 5 f16x3 2>&1 | grep -v amdgpu
done
