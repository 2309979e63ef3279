#!/bin/bash
set -o pipefail
R=/root/repo; O=$R/gpurun_out/qt; mkdir -p $O
cd $R
rc=0

timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dropin-files 300 --aniso-rows 0 --other-rows 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/qt/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ["p50_query_ms", "p50_query_encode_ms", "p50_query_from_tokens_ms", "p50_query_from_text_ms", "p99_query_from_text_ms"]})
PY
