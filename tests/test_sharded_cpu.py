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
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle_engine import OracleEngine
        from voitta_rag_amd.sharded import ShardedSearcher

        x, sp, q, sq = _data()
        mine = np.arange(rank, len(x), world)  # round-robin shard: local row r <-> global r*world+rank
        shard = OracleEngine(x.shape[1])
        shard.upsert(x[mine], sparse=[sp[i] for i in mine])
        s = ShardedSearcher(shard)
        s.replicate_all()  # the shards were filled locally: exchange the term ids once (collection-wide df from here on)
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


# ---- batches and the sharded STORE (ShardedVectorStore), still world-size 2 over gloo ---------------------------

def _store_data():
    from voitta_rag_amd.vector_store import ChunkMetadata

    rng = np.random.default_rng(7)
    dim = 32
    files = [("docs/a.md", "docs", "docs"), ("docs/sub/b.md", "docs/sub", "docs"), ("notes/c.txt", "notes", "notes"),
             ("d.txt", "", ""), ("docs/e.md", "docs", "docs"), ("notes/f.txt", "notes", "notes"), ("x/g.md", "x", "x")]
    chunks, sparse = [], []
    for fi, (fp, folder, ifolder) in enumerate(files):
        n = 6 + fi
        for i in range(n):
            meta = ChunkMetadata(file_path=fp, folder_path=folder, index_folder=ifolder, file_name=fp.split("/")[-1],
                                 chunk_index=i, total_chunks=n, start_char=0, end_char=5, indexed_at="t",
                                 source_modified_at=None if fi == 2 else 1_700_000_000 + 100 * fi + i,
                                 source_url="https://x/doc" if fi == 4 else None, source_page_count=9 if fi == 1 else None)
            chunks.append((f"text {fp} {i}", rng.standard_normal(dim).astype(np.float32).tolist(), meta))
            m = int(rng.integers(1, 7))
            sparse.append(((rng.choice(30, size=m, replace=False) * 11 + 3).astype(int).tolist(),
                           rng.uniform(0.3, 2.0, size=m).astype(np.float32).tolist()))
    queries = [(rng.standard_normal(dim).astype(np.float32).tolist(),
                ((rng.choice(30, size=3, replace=False) * 11 + 3).astype(int).tolist(), [1.0, 1.0, 1.0])) for _ in range(6)]
    return dim, chunks, sparse, queries


def _exercise(vs):
    """The same calls against a plain VectorStoreService and against a ShardedVectorStore: ids aside, answers must
    be equal. Returns a JSON-like trace."""
    dim, chunks, sparse, queries = _store_data()
    key = lambda c: (c.metadata.file_path, c.metadata.chunk_index, c.score, c.text, c.metadata.allowed_users)  # noqa: E731
    trace = []
    ids = []
    for a, b in ((0, 20), (20, 21), (21, len(chunks))):
        ids += vs.store_chunks(chunks[a:b], sparse_vectors=sparse[a:b])
    trace.append(("ids", len(ids), len(set(ids)), all(isinstance(i, str) for i in ids)))
    trace.append(vs.get_collection_info()["points_count"])
    trace.append([vs.count_by_file(f) for f in ("docs/a.md", "x/g.md", "nope")])
    trace.append(vs.count_chunks_for_files(["docs/a.md", "nope", "d.txt", "x/g.md"]))
    trace.append([vs.count_chunks_for_folder(f) for f in ("docs", "", "notes", "zzz")])
    trace.append(vs.get_folder_stats_batch(["docs", "docs/sub", "zzz", "x"]))
    trace.append(sorted(vs.get_file_chunk_counts("docs/").items()))
    trace.append(sorted(vs.get_file_paths_by_index_folder("docs")))
    trace.append([vs.get_stored_page_count("docs/sub/b.md"), vs.get_stored_page_count("docs/a.md")])
    trace.append([key(c) for c in vs.get_chunks_by_range("docs/e.md", 2, 5)])
    trace.append([key(c) for c in vs.find_by_source_url("https://x/doc")])

    def searches():
        out = []
        for qv, sq in queries:
            out.append([key(c) for c in vs.search(qv, limit=7)])
            out.append([key(c) for c in vs.search(qv, limit=5, sparse_query=sq, sparse_weight=0.3)])
            out.append([key(c) for c in vs.search(qv, limit=5, sparse_query=sq, include_folders=["docs", "notes", "ghost"],
                                                  exclude_index_folders=["notes"])])
            out.append([key(c) for c in vs.search(qv, limit=6, folder_filter="docs", date_start=1_700_000_003)])
            out.append(vs.search(qv, limit=0))
        return out

    trace.append(searches())
    trace.append([vs.delete_by_file("docs/a.md"), vs.delete_by_file("docs/a.md"), vs.delete_by_folder("notes"),
                  vs.delete_by_index_folder("x")])
    vs.set_file_acl("d.txt", ["ann@x"])
    trace.append(vs.get_collection_info()["points_count"])
    trace.append(searches())
    return trace


def _store_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EMBEDDING_DIMENSION="32")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle_engine import OracleEngine
        from voitta_rag_amd import config, store_registry
        from voitta_rag_amd.sharded import ShardedSearcher, ShardedVectorStore
        from voitta_rag_amd.vector_store import VectorStoreService

        config.get_settings.cache_clear()
        store_registry.set_engine(OracleEngine(32))
        ret[("store", rank)] = _exercise(ShardedVectorStore(VectorStoreService()))
        # batched searches = the single-query searches, one all_gather / all_reduce per batch
        eng = store_registry.get_engine()
        s = ShardedSearcher(eng)
        _, _, _, queries = _store_data()
        qm = np.array([q for q, _ in queries], np.float32)
        single_d = [s.search_dense(q, 9) for q in qm]
        batch_d = s.search_dense_batch(qm, 9)
        single_h = [s.search_hybrid(q, sq[0], sq[1], 4, 0.3) for q, (_, sq) in zip(qm, queries)]
        batch_h = s.search_hybrid_batch(qm, [sq for _, sq in queries], 4, 0.3)
        same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(single_d, batch_d))
        same &= all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
                    for a, b in zip(single_h, batch_h))
        ret[("batch", rank)] = bool(same) and len(batch_d[0][0]) == 9
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_store_equals_one_store(monkeypatch):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    from voitta_rag_amd import config, store_registry
    from voitta_rag_amd.vector_store import VectorStoreService

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_store_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    monkeypatch.setenv("EMBEDDING_DIMENSION", "32")
    config.get_settings.cache_clear()
    store_registry.set_engine(OracleEngine(32))
    try:
        want = _exercise(VectorStoreService())
    finally:
        store_registry.set_engine(None)
        config.get_settings.cache_clear()
    for rank in range(world):
        got = ret[("store", rank)]
        assert len(got) == len(want)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, (rank, i)
        assert ret[("batch", rank)] is True
