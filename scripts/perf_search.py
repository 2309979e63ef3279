"""Ad-hoc timing of the search path (not the contract bench): N x D corpus generated on device."""
import sys
import time

import numpy as np
import torch

from voitta_rag_amd import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nnz_per = 40
torch.manual_seed(0)
dev = torch.device("cuda:0")
e = Engine(dim, initial_rows=n)
t0 = time.time()
chunk = 100_000
g = torch.Generator(device=dev).manual_seed(1)
for a in range(0, n, chunk):
    b = min(n, a + chunk)
    x = torch.randn((b - a, dim), device=dev, generator=g)
    x = torch.nn.functional.normalize(x, dim=1).contiguous()
    # synthetic sparse rows: Zipf-ish ids, sorted per row
    ids = (torch.rand((b - a, nnz_per), device=dev, generator=g) ** 3 * 200_000).to(torch.int32)
    ids, _ = torch.sort(ids, dim=1)
    # make unique by adding position (keeps sorted, ids stay >= 0)
    ids = ids + torch.arange(nnz_per, device=dev, dtype=torch.int32)[None, :]
    off = (torch.arange(b - a + 1, device=dev, dtype=torch.int64) * nnz_per).contiguous()
    val = torch.rand(((b - a) * nnz_per,), device=dev, generator=g) + 0.5
    e.upsert(x, sparse=(off, ids.reshape(-1).contiguous(), val.contiguous()))
e.sync()
print(f"indexed {n} x {dim} in {time.time()-t0:.2f}s", e.count())
q = torch.nn.functional.normalize(torch.randn((1000, dim), device=dev, generator=g), dim=1).cpu().numpy()
qi = np.array([5, 1000, 20000, 150000, 77], np.int32)
qv = np.ones(5, np.float32)


def timeit(fn, reps):
    fn(0)
    ts = []
    for i in range(reps):
        t = time.perf_counter()
        fn(i)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    return np.percentile(ts, 50), np.percentile(ts, 99), ts.mean()


print("dense top-10 1q   p50/p99/mean ms", timeit(lambda i: e.search_dense(q[i:i + 1], 10), 200))
print("dense top-30 1q   p50/p99/mean ms", timeit(lambda i: e.search_dense(q[i:i + 1], 30), 200))
print("dense top-10 16q  p50/p99/mean ms", timeit(lambda i: e.search_dense(q[16 * i:16 * i + 16], 10), 50))
print("sparse top-30     p50/p99/mean ms", timeit(lambda i: e.search_sparse(qi + i, qv, 30), 200))
print("hybrid top-10     p50/p99/mean ms", timeit(lambda i: e.search_hybrid(q[i], qi + i, qv, 10, 0.1), 200))
print("bytes per dense scan: %.3f GB" % (n * dim * 4 / 1e9))
