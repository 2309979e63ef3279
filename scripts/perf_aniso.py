"""Two-stage dense search on an ANISOTROPIC corpus: rows = common direction + noise, pairwise
cosine ~0.7 — what real sentence-embedding collections look like, where the bound of the int8 prefilter has to
separate scores that sit 3x closer together than on random unit rows. Reports candidates per query, fallbacks and the
single-stream latency for isotropic and anisotropic corpora of the same size, dense and hybrid-free (dense only).
usage: python scripts/perf_aniso.py [rows=1000000] [dim=768] [cos=0.7]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voitta_rag_amd import Engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
cos = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(7)
common = torch.nn.functional.normalize(torch.randn(dim, device=dev, generator=gen), dim=0)


def make(n, aniso):
    u = torch.randn((n, dim), device=dev, generator=gen)
    if aniso:
        u = u - (u @ common)[:, None] * common[None, :]
        u = torch.nn.functional.normalize(u, dim=1)
        u = (cos ** 0.5) * common[None, :] + ((1 - cos) ** 0.5) * u
    return torch.nn.functional.normalize(u, dim=1).contiguous()


for kind in ("isotropic", "anisotropic"):
    e = Engine(dim, initial_rows=rows)
    keep = []
    for a in range(0, rows, 100_000):
        x = make(min(100_000, rows - a), kind == "anisotropic")
        e.upsert(x)
        keep.append(x)
    xs = torch.cat(keep)
    del keep
    for qkind in ("isotropic", "anisotropic"):
        qs = make(320, qkind == "anisotropic")
        qh = qs.cpu().numpy()
        for i in range(20):
            e.search_dense(qh[i:i + 1], 30)
        s0 = e.stats()
        lat, cand, wrong = [], [], 0
        for i in range(20, 320):
            t0 = time.perf_counter()
            got = e.search_dense(qh[i:i + 1], 30)[0]
            lat.append(time.perf_counter() - t0)
            cand.append(e.stats()["last_candidates"])
            if i < 60:  # the same set as an f32 matmul finds (ties aside; the bit-exact checks are in tests/)
                ref = torch.topk(xs @ qs[i], 30).indices.cpu().numpy()
                wrong += len(set(ref.tolist()) ^ set(got[0].tolist()))
        s1 = e.stats()
        print(f"{kind:11s} corpus {rows}x{dim}, {qkind:11s} queries: p50 {np.percentile(lat, 50) * 1e3:.3f} ms  p99 "
              f"{np.percentile(lat, 99) * 1e3:.3f} ms  candidates median {int(np.median(cand))} max {max(cand)}  "
              f"two-stage {s1['two_stage'] - s0['two_stage']}  fallbacks {s1['fallback'] - s0['fallback']}  set mismatches {wrong}", flush=True)
    qb = make(1000, kind == "anisotropic").cpu().numpy()
    e.search_dense(qb, 10)
    s0 = e.stats()
    t0 = time.perf_counter()
    for _ in range(3):
        e.search_dense(qb, 10)
    dt = (time.perf_counter() - t0) / 3
    s1 = e.stats()
    print(f"{kind:11s} corpus, 1000 batched queries of the same kind: {dt * 1e3:.2f} ms per call = {1000 / dt:.0f} QPS, "
          f"batched {s1['batched'] - s0['batched']} fallbacks {s1['batch_fallback'] - s0['batch_fallback']}", flush=True)
    del xs
    e.close()
    torch.cuda.empty_cache()
