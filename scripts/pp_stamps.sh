#!/bin/bash
# In-kernel stamps of gemm_f16_pp_kernel: `make stamps` (the shipped code path + stamps; give 0 as the only switch value) or
# `make diag` with VOITTA_ENGINE_LIB=.../libvoitta_engine_diag.so (one line set per VR_GEMM_DIAG value: 64 no epilogue
# arithmetic, 32 no stores, 1 no loads, 4 no MFMAs, 8 no fragment reads, 16 no barriers)
# usage: scripts/pp_stamps.sh OUT 0            |   VOITTA_ENGINE_LIB=...diag.so scripts/pp_stamps.sh OUT 0 64 32
export PYTHONPATH=/root/repo VOITTA_ENGINE_LIB=${VOITTA_ENGINE_LIB:-/root/repo/voitta_rag_amd/libvoitta_engine_stamps.so}
OUT=$1; shift
: > $OUT
for d in "$@"; do
  echo "=== VR_GEMM_DIAG=$d" >> $OUT
  VR_GEMM_DIAG=$d VR_GEMM_STAMPS=6 timeout -k 10 200 python scripts/perf_encode.py bge-base-en-v1.5 2200 0 1 f16 >> $OUT 2>&1 || exit 1
done
python - $OUT <<'PY'
import re, sys
cur = None; rows = {}
for line in open(sys.argv[1]):
    if line.startswith("==="): diag = line.split("=")[-1].strip()
    m = re.match(r"\[pp stamps\] epi (\d+) M \d+ N (\d+) K (\d+)", line)
    if m: cur = (diag, m.group(1), m.group(2), m.group(3)); rows.setdefault(cur, {0: [], 4: []}); continue
    m = re.match(r"\s+wave (\d) tile\s+(\d+): main\s+(\d+) \(first\s+(-?\d+), second\s+(-?\d+)\) \|\s+(-?\d+) \|\s+(-?\d+) \|\s+(-?\d+)", line)
    if m and cur and int(m.group(2)) >= 1:
        rows[cur][int(m.group(1))].append([int(x) for x in m.groups()[2:]])
print("diag epi N K | wave: main first second align epilogue gap (medians, cycles)")
for k, v in rows.items():
    out = []
    for w in (0, 4):
        if v[w]:
            cols = list(zip(*v[w])); med = [sorted(c)[len(c) // 2] for c in cols]
            out.append("w%d: %6d %5d %5d %4d %6d %5d" % (w, *med))
    print(" ".join(k), "|", " | ".join(out))
PY
