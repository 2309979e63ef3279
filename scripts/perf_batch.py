"""Batched dense search alone (the two integer-GEMM passes of csrc/batch.hip), for rocprofv3 --pmc passes.
usage: python scripts/perf_batch.py [rows=200000] [queries=1000] [calls=3]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voitta_rag_amd import Engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dim = 768
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(7)
e = Engine(dim, initial_rows=rows + 64)
for a in range(0, rows, 50_000):
    n = min(50_000, rows - a)
    e.upsert(torch.nn.functional.normalize(torch.randn((n, dim), device=dev, generator=gen), dim=1))
q = torch.nn.functional.normalize(torch.randn((nq, dim), device=dev, generator=gen), dim=1).cpu().numpy()
e.search_dense(q, 10)
e.profile(True)
t0 = time.perf_counter()
for _ in range(calls):
    e.search_dense(q, 10)
dt = (time.perf_counter() - t0) / calls
ms, n, w = e.profile_read(Engine.PROF_BATCH_SCAN)
print(f"{rows} rows, {nq} queries: {dt * 1e3:.3f} ms per call = {nq / dt:.0f} QPS; scan kernels {ms / max(n, 1):.3f} ms per call "
      f"({2 * w / max(ms, 1e-9) / 1e9:.0f} TOP/s executed on {2 * rows * dim * nq * 2 / 1e12:.2f} TOP per pass)", e.stats())
e.close()
