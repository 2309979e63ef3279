"""Few hybrid queries for a rocprofv3 timeline (kernel + memcpy trace)."""
import sys, time
import numpy as np, torch
from voitta_rag_amd import Engine
n, dim, nnz_per = 1_000_000, 768, 40
dev = torch.device("cuda:0")
e = Engine(dim, initial_rows=n)
g = torch.Generator(device=dev).manual_seed(1)
for a in range(0, n, 100_000):
    x = torch.nn.functional.normalize(torch.randn((100_000, dim), device=dev, generator=g), dim=1).contiguous()
    ids = (torch.rand((100_000, nnz_per), device=dev, generator=g) ** 3 * 200_000).to(torch.int32)
    ids, _ = torch.sort(ids, dim=1)
    ids = ids + torch.arange(nnz_per, device=dev, dtype=torch.int32)[None, :]
    off = (torch.arange(100_001, device=dev, dtype=torch.int64) * nnz_per).contiguous()
    val = torch.rand((100_000 * nnz_per,), device=dev, generator=g) + 0.5
    e.upsert(x, sparse=(off, ids.reshape(-1).contiguous(), val.contiguous()))
e.sync()
q = torch.nn.functional.normalize(torch.randn((64, dim), device=dev, generator=g), dim=1).cpu().numpy()
qi = np.array([5, 1000, 20000, 150000, 77], np.int32); qv = np.ones(5, np.float32)
for i in range(10): e.search_hybrid(q[i], qi + i, qv, 10, 0.1)
torch.cuda.synchronize(); time.sleep(0.05)
print("MARK", time.time_ns())
for i in range(10, 40):
    t = time.perf_counter(); e.search_hybrid(q[i], qi + i, qv, 10, 0.1); print("wall_us", (time.perf_counter() - t) * 1e6)
