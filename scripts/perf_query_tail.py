"""Where the single-query hybrid latency goes (bench.py's corpus and queries): p50 of each leg alone and together.
usage: python scripts/perf_query_tail.py [rows=1000000]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from voitta_rag_amd import Engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = 768
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1234)
e = Engine(dim, initial_rows=rows + 64)
bench.populate(torch, gen, dev, e, rows, dim)
qgen = torch.Generator(device=dev).manual_seed(99)
nq = 520
qs = torch.nn.functional.normalize(torch.randn((nq, dim), device=dev, generator=qgen), dim=1).cpu().numpy()
terms = bench.stem_hash(torch, bench.zipf_ids(qgen, torch, nq * 6, 30000, dev)).view(-1, 6).cpu().numpy()
nnz = np.random.default_rng(5).integers(4, 7, size=nq)
ones = np.ones(8, np.float32)


def p50(fn):
    for i in range(20):
        fn(i)
    lat = np.empty(500)
    for i in range(500):
        t0 = time.perf_counter()
        fn(20 + i)
        lat[i] = time.perf_counter() - t0
    return f"p50 {np.percentile(lat, 50) * 1e3:.3f} ms  p99 {np.percentile(lat, 99) * 1e3:.3f} ms"


print("dense top-10            ", p50(lambda i: e.search_dense(qs[i:i + 1], 10)))
print("dense top-30            ", p50(lambda i: e.search_dense(qs[i:i + 1], 30)))
print("sparse top-30           ", p50(lambda i: e.search_sparse(terms[i, :nnz[i]], ones[:nnz[i]], 30)))
print("hybrid, no sparse terms ", p50(lambda i: e.search_hybrid(qs[i], terms[i, :0], ones[:0], 10, 0.1)))
print("hybrid top-10           ", p50(lambda i: e.search_hybrid(qs[i], terms[i, :nnz[i]], ones[:nnz[i]], 10, 0.1)))
os.environ["X"] = "1"
e.profile(True)
for i in range(200):
    e.search_hybrid(qs[i], terms[i, :nnz[i]], ones[:nnz[i]], 10, 0.1)
for name, cls in (("dense scan", Engine.PROF_DENSE_SCAN), ("sparse scan", Engine.PROF_SPARSE_SCAN)):
    ms, n, w = e.profile_read(cls)
    print(f"{name}: {ms / max(n, 1) * 1e3:.1f} us per launch, {w / max(ms, 1e-9) / 1e6:.0f} GB/s")
e.close()
