"""Where a batched hybrid call spends its time: python scripts/perf_hybrid_batch.py [rows] [queries] [reps]
(bench.py's corpus: unit rows x 768 + 40-term Zipf BM25 rows; queries of 4-6 Zipf terms)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from voitta_rag_amd import Engine  # noqa: E402
from voitta_rag_amd.engine import fuse_batch  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sparse_only = len(sys.argv) > 4 and sys.argv[4] == "sparse_only"   # (timing experiments on the sparse batch kernels)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
dim = 768
e = Engine(dim, initial_rows=rows + 64)
bench.populate(torch, gen, dev, e, rows, dim)
qgen = torch.Generator(device=dev).manual_seed(99)
qs = torch.nn.functional.normalize(torch.randn((nq, dim), device=dev, generator=qgen), dim=1).cpu().numpy()
q_terms = bench.stem_hash(torch, bench.zipf_ids(qgen, torch, nq * 6, 30000, dev)).view(-1, 6).cpu().numpy()
q_nnz = np.random.default_rng(5).integers(4, 7, size=nq)
ones = np.ones(8, np.float32)
sq = [(q_terms[i, : q_nnz[i]], ones[: q_nnz[i]]) for i in range(nq)]
sq_off = np.zeros(nq + 1, np.int64)
sq_off[1:] = np.cumsum(q_nnz)
sq_csr = (sq_off, np.concatenate([q[0] for q in sq]).astype(np.int32), np.ones(int(sq_off[-1]), np.float32))


def timed(name, fn, n=reps):
    fn()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    e.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:42s} {dt * 1e3:9.3f} ms per call", flush=True)
    return out


for kk in (() if sparse_only else (10, 30)):
    s0 = e.stats()
    timed(f"dense batch k={kk} (raw arrays)", lambda: e.search_dense(qs, kk, raw=True))
    s1 = e.stats()
    print(f"  candidates re-scored per query: {(s1['batch_candidates'] - s0['batch_candidates']) / max(s1['batched'] - s0['batched'], 1):.1f}, "
          f"fallbacks {s1['batch_fallback'] - s0['batch_fallback']}")
e.profile(True)
sp = timed("sparse batch k=30", lambda: e.search_sparse_batch(sq, 30))
ms, n, _ = e.profile_read(Engine.PROF_SPARSE_SCAN)
print(f"  sparse_inv_batch_kernel: {ms / max(n, 1):.3f} ms per launch ({n} launches)")
e.profile(False)
if sparse_only:
    st = e.stats()
    e.search_sparse_batch(sq[:16], 30)  # (the candidate count of a batch is read when the next one starts)
    st = e.stats()
    print(f"  grouped queries {st.get('sparse_grouped')}, batches redone {st.get('sparse_group_redo')}, "
          f"candidates ranked {st.get('sparse_group_candidates')}")
    e.close()
    sys.exit(0)
timed("hybrid batch limit=10", lambda: e.search_hybrid_batch(qs, sq, 10, 0.1, raw=True))
timed("hybrid batch limit=10 (CSR in, raw out)", lambda: e.search_hybrid_batch(qs, sq_csr, 10, 0.1, raw=True))
keys = timed("hybrid keys k=30", lambda: e.search_hybrid_keys(qs, sq, 30))
g, s, c = timed("merge_keys (1 part)", lambda: e.merge_keys(keys[None], 30))
g, s, c = g.reshape(nq, 2, 30), s.reshape(nq, 2, 30), c.reshape(nq, 2)
timed("fuse_batch", lambda: fuse_batch(g[:, 0], s[:, 0], c[:, 0], g[:, 1], s[:, 1], c[:, 1], 10, 0.1))
timed("100 single sparse k=30", lambda: [e.search_sparse(sq[i][0], sq[i][1], 30) for i in range(100)], 2)
timed("100 single hybrid limit=10", lambda: [e.search_hybrid(qs[i], sq[i][0], sq[i][1], 10, 0.1) for i in range(100)], 2)
timed("1 hybrid keys (nq=1) x100", lambda: [e.search_hybrid_keys(qs[i:i + 1], sq[i:i + 1], 30) for i in range(100)], 2)
e.close()
