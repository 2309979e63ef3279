"""Search-path timing: host wall p50 plus the engine's own HIP-event kernel times."""
import sys
import time

import numpy as np
import torch

from voitta_rag_amd import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nnz_per = 40
dev = torch.device("cuda:0")
e = Engine(dim, initial_rows=n)
g = torch.Generator(device=dev).manual_seed(1)
for a in range(0, n, 100_000):
    b = min(n, a + 100_000)
    x = torch.nn.functional.normalize(torch.randn((b - a, dim), device=dev, generator=g), dim=1).contiguous()
    ids = (torch.rand((b - a, nnz_per), device=dev, generator=g) ** 3 * 200_000).to(torch.int32)
    ids, _ = torch.sort(ids, dim=1)
    ids = ids + torch.arange(nnz_per, device=dev, dtype=torch.int32)[None, :]
    off = (torch.arange(b - a + 1, device=dev, dtype=torch.int64) * nnz_per).contiguous()
    val = torch.rand(((b - a) * nnz_per,), device=dev, generator=g) + 0.5
    e.upsert(x, sparse=(off, ids.reshape(-1).contiguous(), val.contiguous()))
e.sync()
q = torch.nn.functional.normalize(torch.randn((1000, dim), device=dev, generator=g), dim=1).cpu().numpy()
qi = np.array([5, 1000, 20000, 150000, 77], np.int32)
qv = np.ones(5, np.float32)


def run(name, fn, reps):
    fn(0)
    e.profile(True)
    ts = []
    for i in range(reps):
        t = time.perf_counter()
        fn(i)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    d = e.profile_read(Engine.PROF_DENSE_SCAN)
    s = e.profile_read(Engine.PROF_SPARSE_SCAN)
    e.profile(False)
    msg = f"{name:22s} wall p50 {np.percentile(ts,50):.4f} ms p99 {np.percentile(ts,99):.4f}"
    if d[1]:
        msg += f" | dense scan {d[0]/d[1]*1e3:.1f} us/launch {d[2]/d[0]/1e6:.0f} GB/s"
    if s[1]:
        msg += f" | sparse scan {s[0]/s[1]*1e3:.1f} us/launch {s[2]/s[0]/1e6:.0f} GB/s"
    print(msg)


run("dense top-10 1q", lambda i: e.search_dense(q[i:i + 1], 10), 200)
run("dense top-30 1q", lambda i: e.search_dense(q[i:i + 1], 30), 200)
run("dense top-100 1q (old)", lambda i: e.search_dense(q[i:i + 1], 100), 100)
run("dense top-10 16q", lambda i: e.search_dense(q[16 * i:16 * i + 16], 10), 50)
run("sparse top-30", lambda i: e.search_sparse(qi + i, qv, 30), 200)
run("hybrid top-10", lambda i: e.search_hybrid(q[i], qi + i, qv, 10, 0.1), 200)
