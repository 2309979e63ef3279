"""Multi-GPU path with REAL engines: two gloo ranks that share GPU 0 (the driver's 8-GPU run is the first time RCCL
itself is initialised; here everything else of that path runs on hardware). Each rank holds a shard behind the real
VectorStoreService / Engine; the sharded store must answer exactly as ONE store whose engine is the CPU oracle
(tests/oracle_engine.py) holding the whole corpus — the same trace as tests/test_sharded_cpu.py checks on CPU."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from test_sharded_cpu import _exercise, _free_port, _store_data  # noqa: E402

pytestmark = pytest.mark.gpu


def _gpu_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EMBEDDING_DIMENSION="32", VOITTA_GPU="0")
    os.environ.pop("VOITTA_INDEX_DIR", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, HERE)
        from voitta_rag_amd import Engine, config, store_registry
        from voitta_rag_amd.sharded import ShardedSearcher, ShardedVectorStore
        from voitta_rag_amd.vector_store import VectorStoreService

        config.get_settings.cache_clear()
        store_registry.reset()
        ret[("store", rank)] = _exercise(ShardedVectorStore(VectorStoreService()))
        assert isinstance(store_registry.get_engine(), Engine)  # the real thing, not a double
        # a bigger shard pair for the searcher: batches == single queries == what the keys API returns
        rng = np.random.default_rng(5)
        n, dim = 40000, 128  # 20000 rows per shard: above the batched search's threshold
        x = rng.standard_normal((n, dim)).astype(np.float32)
        e = Engine(dim, initial_rows=n)
        mine = np.arange(rank, n, world)
        sp = [((rng.choice(300, size=5, replace=False) * 7 + 1).astype(np.int32), rng.uniform(0.5, 2.0, size=5).astype(np.float32))
              for _ in range(n)]
        e.upsert(x[mine], sparse=[sp[i] for i in mine])
        s = ShardedSearcher(e)
        s.replicate_all()  # filled locally: one exchange of term ids makes every shard's df table collection-wide
        q = rng.standard_normal((40, dim)).astype(np.float32)
        sq = [((rng.choice(300, size=3, replace=False) * 7 + 1).astype(np.int32), np.ones(3, np.float32)) for _ in range(40)]
        batch_d = s.search_dense_batch(q, 10)       # > 16 queries: the integer-GEMM batched search on every shard
        batch_h = s.search_hybrid_batch(q, sq, 5, 0.25)
        ok = e.stats()["batched"] >= 40
        for i in range(0, 40, 3):
            d = s.search_dense(q[i], 10)
            h = s.search_hybrid(q[i], sq[i][0], sq[i][1], 5, 0.25)
            ok &= np.array_equal(d[0], batch_d[i][0]) and np.array_equal(d[1], batch_d[i][1])
            ok &= all(np.array_equal(a, b) for a, b in zip(h, batch_h[i]))
        ret[("batch", rank)] = (bool(ok), [(g.tolist(), sc.tolist()) for g, sc in batch_d[:8]],
                                [(g.tolist(), sc.tolist()) for g, sc, _ in batch_h[:4]])
        e.close()
        store_registry.reset()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_with_real_engines_equal_one_oracle_store(gpu, monkeypatch):
    from oracle import core as ocore
    from oracle_engine import OracleEngine
    from voitta_rag_amd import config, store_registry
    from voitta_rag_amd.vector_store import VectorStoreService

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    monkeypatch.setenv("EMBEDDING_DIMENSION", "32")
    config.get_settings.cache_clear()
    store_registry.set_engine(OracleEngine(32))
    try:
        want = _exercise(VectorStoreService())
    finally:
        store_registry.set_engine(None)
        config.get_settings.cache_clear()
    # the merged dense answers of the big shard pair against one oracle scan of the whole corpus
    rng = np.random.default_rng(5)
    n, dim = 40000, 128
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sp = [((rng.choice(300, size=5, replace=False) * 7 + 1).astype(np.int32), rng.uniform(0.5, 2.0, size=5).astype(np.float32))
          for _ in range(n)]
    q = rng.standard_normal((40, dim)).astype(np.float32)
    sq = [((rng.choice(300, size=3, replace=False) * 7 + 1).astype(np.int32), np.ones(3, np.float32)) for _ in range(40)]
    sc = ocore.dense_scores(ocore.cosine_preprocess(q[:8]), ocore.cosine_preprocess(x))
    # the merged HYBRID answers against one oracle over the whole corpus (collection-wide IDF, fusion after the merge)
    whole = OracleEngine(dim)
    whole.upsert(x, sparse=sp)
    want_h = [whole.search_hybrid(q[i], sq[i][0], sq[i][1], 5, 0.25) for i in range(4)]
    for rank in range(world):
        got = ret[("store", rank)]
        assert len(got) == len(want)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, (rank, i)
        ok, first, first_h = ret[("batch", rank)]
        assert ok
        for i in range(4):
            assert first_h[i][0] == want_h[i][0].tolist() and first_h[i][1] == want_h[i][1].tolist(), (rank, i)
        for i in range(8):  # round-robin sharding: global id == original row
            wr, ws = ocore.topk(sc[i], 10)
            assert first[i][0] == wr.tolist() and first[i][1] == ws.tolist()


def _rccl_worker(rank, world, port, ret):
    """world = 1 over the REAL nccl (= RCCL) backend: the device-resident code path of ShardedSearcher — engine keys
    written into a device tensor, all_gather_into_tensor on it, the device-side merge — on the one GPU a box has."""
    import torch

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from voitta_rag_amd import Engine
        from voitta_rag_amd.sharded import ShardedSearcher

        rng = np.random.default_rng(9)
        n, dim = 20000, 128
        x = rng.standard_normal((n, dim)).astype(np.float32)
        sp = [((rng.choice(300, size=5, replace=False) * 7 + 1).astype(np.int32), rng.uniform(0.5, 2.0, size=5).astype(np.float32))
              for _ in range(n)]
        e = Engine(dim, initial_rows=n)
        e.upsert(x, sparse=sp)
        s = ShardedSearcher(e)
        assert s.on_device  # nccl: the communication tensors live on the GPU
        s.replicate_all()   # (one rank: nothing to apply, but the device-side export and both collectives run)
        q = rng.standard_normal((40, dim)).astype(np.float32)
        sq = [((rng.choice(300, size=3, replace=False) * 7 + 1).astype(np.int32), np.ones(3, np.float32)) for _ in range(40)]
        batch_d = s.search_dense_batch(q, 10)
        batch_h = s.search_hybrid_batch(q, sq, 5, 0.25)
        ok = True
        for i in range(40):
            rows, scores = e.search_dense(q[i:i + 1], 10)[0]   # world = 1: global id == row
            ok &= np.array_equal(batch_d[i][0], rows) and np.array_equal(batch_d[i][1].view(np.uint32), scores.view(np.uint32))
            d1 = s.search_dense(q[i], 10)
            ok &= np.array_equal(d1[0], rows) and np.array_equal(d1[1].view(np.uint32), scores.view(np.uint32))
            hr, hs, hf = e.search_hybrid(q[i], sq[i][0], sq[i][1], 5, 0.25)
            bh = batch_h[i]
            ok &= np.array_equal(bh[0], hr) and np.array_equal(bh[1], hs)
            h1 = s.search_hybrid(q[i], sq[i][0], sq[i][1], 5, 0.25)
            ok &= np.array_equal(h1[0], hr) and np.array_equal(h1[1], hs)
        ret["ok"] = bool(ok)
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_device_resident_merge_over_rccl_with_one_rank(gpu):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret.get("ok") is True
