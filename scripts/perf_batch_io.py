"""Batched dense search: host arrays in vs device tensor in (what the H2D of 1000 x 768 queries costs a call).
usage: python scripts/perf_batch_io.py [rows=1000000] [queries=1000]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voitta_rag_amd import Engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dim = 768
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(7)
e = Engine(dim, initial_rows=rows + 64)
for a in range(0, rows, 100_000):
    n = min(100_000, rows - a)
    e.upsert(torch.nn.functional.normalize(torch.randn((n, dim), device=dev, generator=gen), dim=1))
qd = torch.nn.functional.normalize(torch.randn((nq, dim), device=dev, generator=gen), dim=1).contiguous()
qh = qd.cpu().numpy()
qp = torch.from_numpy(qh).pin_memory().numpy()


def timed(name, fn, n=10):
    fn(); e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    e.sync()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms per call", flush=True)


for k in (10, 30):
    timed(f"k={k}: host array in (pageable)", lambda: e.search_dense(qh, k, raw=True))
    timed(f"k={k}: host array in (pinned)", lambda: e.search_dense(qp, k, raw=True))
    timed(f"k={k}: device tensor in", lambda: e.search_dense(qd, k, raw=True))
e.close()
