"""N > 1 path on CPU: world_size-2 gloo processes, each holding one shard behind an oracle-backed
stand-in for the engine; the merged results must equal one oracle over the whole corpus."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import core as ocore
from oracle import fusion as ofus


class OracleShard:
    """Test double with the Engine methods ShardedSearcher uses (test infrastructure only)."""

    def __init__(self, x, sparse_rows):
        self.xh = ocore.cosine_preprocess(x)
        self.sp = sparse_rows

    def search_dense(self, q, k, flt=None):
        sc = ocore.dense_scores(ocore.cosine_preprocess(q), self.xh)
        return [ocore.topk(sc[i], k) for i in range(q.shape[0])]

    def sparse_stats(self, ids):
        df, n = ocore.document_frequencies(self.sp)
        return np.array([df.get(int(t), 0) for t in ids], np.int32), n

    def idf(self, n, df):
        return ocore.idf(n, df)

    def search_sparse(self, ids, w, k, flt=None, weights_given=False):
        assert weights_given
        off, idx, val = ocore.to_csr(self.sp)
        sc = np.full(len(self.sp), -np.inf, np.float32)
        for r in range(len(self.sp)):
            acc, hit = np.float32(0), False
            for j in range(off[r], off[r + 1]):
                m = np.nonzero(ids == idx[j])[0]
                if len(m):
                    acc = np.float32(acc + np.float32(w[m[0]] * val[j]))
                    hit = True
            if hit:
                sc[r] = acc
        return ocore.topk(sc, k)


def _data():
    rng = np.random.default_rng(42)
    n, dim = 301, 32
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sp = []
    for _ in range(n):
        m = int(rng.integers(0, 12))
        ids = rng.choice(60, size=m, replace=False).astype(np.int32) * 101 + 7
        sp.append((ids, rng.uniform(0.3, 2.0, size=m).astype(np.float32)))
    q = rng.standard_normal((4, dim)).astype(np.float32)
    sq = [(rng.choice(60, size=3, replace=False).astype(np.int32) * 101 + 7, np.ones(3, np.float32)) for _ in range(4)]
    return x, sp, q, sq


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from voitta_rag_amd.sharded import ShardedSearcher

        x, sp, q, sq = _data()
        mine = np.arange(rank, len(x), world)  # round-robin shard: local row r <-> global r*world+rank
        s = ShardedSearcher(OracleShard(x[mine], [sp[i] for i in mine]))
        out = []
        for i in range(len(q)):
            d = s.search_dense(q[i], 10)
            sres = s.search_sparse(sq[i][0], sq[i][1], 10)
            h = s.search_hybrid(q[i], sq[i][0], sq[i][1], 5, 0.3)
            out.append((d, sres, h))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
def test_two_rank_merge_equals_single_oracle():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    x, sp, q, sq = _data()
    xh = ocore.cosine_preprocess(x)
    for i in range(len(q)):
        dsc = ocore.dense_scores(ocore.cosine_preprocess(q[i:i + 1]), xh)[0]
        wr, ws = ocore.topk(dsc, 10)
        ssc = ocore.sparse_scores(sp, sq[i][0], sq[i][1])
        sr, ss = ocore.topk(ssc, 10)
        dr30, ds30 = ocore.topk(dsc, 15)
        sr30, ss30 = ocore.topk(ssc, 15)
        fused = ofus.hybrid_fuse(list(zip(dr30.tolist(), ds30.tolist())), list(zip(sr30.tolist(), ss30.tolist())), 5, 0.3)
        for rank in range(world):
            (gd, gds), (gs, gss), (hr, hs, hf) = ret[rank][i]
            # round-robin sharding makes global id == original row
            assert gd.tolist() == wr.tolist() and np.array_equal(gds, ws)
            assert gs.tolist() == sr.tolist() and np.array_equal(gss, ss)
            assert hr.tolist() == [r for r, _, _ in fused]
            assert hs.tolist() == [s for _, s, _ in fused]


def test_shard_of_is_stable():
    from voitta_rag_amd.sharded import shard_of

    assert [shard_of("docs/a.md", 8), shard_of("docs/b.md", 8)] == [shard_of("docs/a.md", 8), shard_of("docs/b.md", 8)]
    assert all(0 <= shard_of(f"f{i}", 8) < 8 for i in range(100))
