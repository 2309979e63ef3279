"""Print the headline numbers of gpurun_out/round/bench_n1.json and the per-launch GEMM time from
the rocprof kernel stats of the same pass (they must agree: profiles/README.md)."""
import csv, glob, json, os
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "round")
b = json.load(open(os.path.join(R, "bench_n1.json")))
print({k: b[k] for k in ("value", "ms_per_step", "p50_query_ms", "p99_query_ms")})
print("roofline", {k: b["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "share_of_step_time", "traffic")})
print("search", {k: b["roofline_search"][k] for k in ("achieved", "frac", "avg_launch_ms", "traffic")})
print("attention", {k: b["roofline_attention"][k] for k in ("achieved", "frac", "avg_launch_ms", "TFLOPs")})
print("batched", b["qps_batched_1k"], {k: b["roofline_batched_search"][k] for k in ("achieved", "frac", "avg_launch_ms")})
print("drop-in", b["dropin_index_chunks_per_s"], "chunks/s;", b["p50_query_from_text_ms"], "ms from text; cpu",
      b["cpu_baseline"]["value"], b["speedup_vs_cpu_index"], b["speedup_vs_cpu_query"])
f = sorted(glob.glob(os.path.join(R, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
g = [r for r in rows if "gemm_f16_pp_kernel" in r["Name"]]
tot, n = sum(float(r["TotalDurationNs"]) for r in g) / 1e6, sum(int(r["Calls"]) for r in g)
print(f"rocprof GEMM: {tot:.2f} ms / {n} launches = {tot / max(n, 1):.4f} ms  ({os.path.basename(f)})")
for tag in ("attention_seq_kernel", "prefilter_scan8_kernel", "batch_scan_kernel"):
    k = [r for r in rows if tag in r["Name"]]
    t, c = sum(float(r["TotalDurationNs"]) for r in k) / 1e6, sum(int(r["Calls"]) for r in k)
    print(f"rocprof {tag}: {t:.2f} ms / {c} launches = {t / max(c, 1):.4f} ms")
for r in rows[:8]:
    print("  %-58s calls %5s avg_us %8.1f" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3))
