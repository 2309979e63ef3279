#!/bin/bash
# One GPU-box call: the whole -m gpu suite, the N=1 bench line, and the 2-rank rehearsal of the N>1 path on one shared GPU.
# Outputs under gpurun_out/check/.
set -o pipefail
R=/root/repo; O=$R/gpurun_out/check; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee $O/gpu_tests.rc; tail -5 $O/gpu_tests.log
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -c 1500 $O/bench_n1.err; python - <<'PY'
import json
try:
    d = json.loads(open('/root/repo/gpurun_out/check/bench_n1.json').read().strip().splitlines()[-1])
    keys = ["value","ms_per_step","p50_query_ms","p99_query_ms","qps_batched_1k","ms_per_batched_call","qps_hybrid_batched_1k","ms_per_hybrid_batched_call",
            "recall_at_10_hybrid_batched_vs_single_query","dropin_index_chunks_per_s","dropin_sync_index_chunks_per_s","p50_query_from_text_ms","other_model_widths"]
    print({k: d.get(k) for k in keys})
    print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "batched", d["roofline_batched_search"]["frac"], "attn", d["roofline_attention"]["frac"])
except Exception as e:
    print("no bench line:", e)
PY
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --corpus 300000 --steps 3 --warmup 1 --queries 300 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "rehearsal rc=$?"; tail -c 600 $O/bench_n2_gloo.err; python - <<'PY'
import json
try:
    d = json.loads(open('/root/repo/gpurun_out/check/bench_n2_gloo.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ["n_gpus","value","p50_query_ms","qps_batched_1k","qps_hybrid_batched_1k","recall_at_10_hybrid_batched_vs_single_query"]})
except Exception as e:
    print("no rehearsal line:", e)
PY
