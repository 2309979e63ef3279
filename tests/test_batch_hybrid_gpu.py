"""Batched sparse / hybrid search (vr_search_sparse_batch, vr_search_hybrid_batch, vr_search_hybrid_keys,
vr_merge_keys) through the C-ABI: every query of a batch must come out bit for bit as the CPU oracle ranks it and as
the single-query entry points return it. Reference behaviour restated: src/voitta/services/vector_store.py:647-656
(sparse query_points), :621-697 (_hybrid_search: prefetch 3 x limit, min-max fusion), caller mcp_server.py:469-485;
BASELINE configs[4] names the batched form ("1k batched queries")."""
import numpy as np
import pytest

from oracle import core as ocore
from oracle import fusion as ofus

pytestmark = pytest.mark.gpu


def _sparse_rows(rng, n, vocab, lo=0, hi=40):
    rows = []
    for _ in range(n):
        m = int(rng.integers(lo, hi + 1))
        ids = np.sort(rng.choice(vocab, size=m, replace=False)).astype(np.int32) * 7919 + 13
        rows.append((ids, rng.uniform(0.2, 2.2, size=m).astype(np.float32)))
    return rows


def _queries(rng, nq, vocab, max_terms=9):
    out = []
    for i in range(nq):
        m = int(rng.integers(0, max_terms + 1))
        if i % 13 == 5:
            m = 0                                        # a query without sparse terms
        if i % 17 == 3:
            m = 40                                       # more than the postings kernel takes: served alone
        ids = (rng.choice(vocab, size=m, replace=False) if m <= vocab else rng.integers(0, vocab, size=m)).astype(np.int32) * 7919 + 13
        vals = rng.uniform(0.5, 1.5, size=m).astype(np.float32)
        if m >= 3 and i % 5 == 0:
            ids[2] = ids[0]                              # a repeated term: the first value counts
        out.append((ids, vals))
    return out


@pytest.fixture(scope="module")
def small(gpu):
    """20,000 rows x 128 (above the 16,384 rows the integer-GEMM dense batch needs), several upsert batches, deletes,
    folders; the oracle holds everything."""
    from voitta_rag_amd import Engine

    rng = np.random.default_rng(77)
    n, dim, vocab = 20_000, 128, 600
    x = rng.standard_normal((n, dim)).astype(np.float32)
    sp = _sparse_rows(rng, n, vocab)
    folder = rng.integers(0, 6, size=n).astype(np.int32)
    e = Engine(dim)
    for a, b in ((0, 7000), (7000, 7013), (7013, 16000), (16000, n)):
        e.upsert(x[a:b], sparse=sp[a:b], folder_ids=folder[a:b])
    dead = rng.choice(n, size=700, replace=False)
    e.delete_rows(dead)
    live = np.ones(n, np.uint8)
    live[dead] = 0
    yield e, rng, x, ocore.SparseOracle(sp, live), folder, live, vocab
    e.close()


def test_sparse_batch_against_the_oracle_and_the_single_search(small):
    from voitta_rag_amd import SearchFilter

    e, rng, x, sp, folder, live, vocab = small
    qs = _queries(rng, 120, vocab)
    for flt, mask in ((None, live.astype(bool)), (SearchFilter(include_folders=[1, 4]), live.astype(bool) & np.isin(folder, [1, 4]))):
        for k in (10, 30, 64, 70):                       # 70: beyond the fused lists -> every query served alone
            got = e.search_sparse_batch(qs, k, flt)
            assert len(got) == len(qs)
            for i, (qi, qv) in enumerate(qs):
                if len(qi) == 0:
                    assert len(got[i][0]) == 0
                    continue
                wr, ws = ocore.topk(sp.scores(qi, qv), k, mask.astype(np.uint8))
                assert np.array_equal(got[i][0], wr), (k, i)
                assert np.array_equal(got[i][1].view(np.uint32), ws.view(np.uint32)), (k, i)
                if i % 7 == 0:
                    r1, s1 = e.search_sparse(qi, qv, k, flt)
                    assert np.array_equal(got[i][0], r1) and np.array_equal(got[i][1].view(np.uint32), s1.view(np.uint32))


def test_sparse_batch_pruning_keeps_the_exact_answer(small, monkeypatch):
    """One block per query walks ALL segments (VR_SPARSE_BATCH_BLOCKS=1): after the first segment its lists are full and
    the others are scanned with dynamic pruning (csrc/invert.hip: only the essential terms' postings, candidates scored
    from the forward index). Rows, scores and order must still be the oracle's, bit for bit — common terms (every row
    has them: pure non-essential ballast), rare terms, a filter, and k from 1 to 64."""
    from voitta_rag_amd import SearchFilter

    e, rng, x, sp, folder, live, vocab = small
    df = sp.df
    common = sorted(df, key=lambda t: -df[t])[:5]
    rare = sorted(df, key=lambda t: df[t])[:200]
    qs = []
    for i in range(80):
        terms = list(rng.choice(common, size=int(rng.integers(0, 3)), replace=False)) + list(rng.choice(rare, size=int(rng.integers(0, 4)), replace=False))
        if not terms:
            terms = [common[0]]
        qs.append((np.array(terms, np.int32), rng.uniform(0.5, 1.5, size=len(terms)).astype(np.float32)))
    for blocks in ("1", "2"):
        monkeypatch.setenv("VR_SPARSE_BATCH_BLOCKS", blocks)
        for flt, mask in ((None, live.astype(bool)), (SearchFilter(exclude_folders=[3]), live.astype(bool) & (folder != 3))):
            for k in (1, 10, 30, 64):
                got = e.search_sparse_batch(qs, k, flt)
                for i, (qi, qv) in enumerate(qs):
                    wr, ws = ocore.topk(sp.scores(qi, qv), k, mask.astype(np.uint8))
                    assert np.array_equal(got[i][0], wr), (blocks, k, i)
                    assert np.array_equal(got[i][1].view(np.uint32), ws.view(np.uint32)), (blocks, k, i)


def test_grouped_sparse_batch_keeps_the_exact_answer(small, monkeypatch):
    """Batches of 16 queries or more take the GROUPED scan (csrc/invert.hip, sparse_inv_group_kernel): groups of up to 8
    queries share a block per segment, a term's postings are read once per group and added to every member's
    accumulators — in ascending term order per query, so the bits stay the forward scan's. Queries from a Zipfian
    vocabulary (shared common terms, the case it is built for), groups of 8 and of 4, a filter, k from 1 to 64; held
    against the oracle and against the per-query kernels (VR_SPARSE_GROUPED=0). Candidates that do not fit
    their (query, segment) region spill into the query's spill area (VR_SPARSE_GROUP_CAP); a spill area that overflows
    (VR_SPARSE_GROUP_SPILL) makes the engine redo THAT query on the per-query kernels: same answer, counted."""
    from voitta_rag_amd import SearchFilter

    e, rng, x, sp, folder, live, vocab = small
    df = sp.df
    by_df = np.array(sorted(df, key=lambda t: -df[t]), np.int32)
    p = 1.0 / np.arange(1, len(by_df) + 1) ** 1.1
    p /= p.sum()
    qs = []
    for i in range(130):
        m = int(rng.integers(1, 9))
        ids = rng.choice(by_df, size=m, replace=False, p=p).astype(np.int32)
        qs.append((ids, rng.uniform(0.5, 1.5, size=m).astype(np.float32)))
    qs[7] = (np.zeros(0, np.int32), np.zeros(0, np.float32))            # no terms
    qs[11] = (np.array([5], np.int32), np.ones(1, np.float32))         # a term no row carries
    for group in ("2", "3", "4", "8"):
        monkeypatch.setenv("VR_SPARSE_GROUP", group)
        for flt, mask in ((None, live.astype(bool)), (SearchFilter(include_folders=[0, 2, 5]), live.astype(bool) & np.isin(folder, [0, 2, 5]))):
            for k in (1, 10, 30, 64):
                before = e.stats()
                got = e.search_sparse_batch(qs, k, flt)
                after = e.stats()
                assert after["sparse_grouped"] - before["sparse_grouped"] == len(qs)
                assert after["sparse_group_redo"] == before["sparse_group_redo"]
                for i, (qi, qv) in enumerate(qs):
                    if len(qi) == 0:
                        assert len(got[i][0]) == 0
                        continue
                    wr, ws = ocore.topk(sp.scores(qi, qv), k, mask.astype(np.uint8))
                    assert np.array_equal(got[i][0], wr), (group, k, i)
                    assert np.array_equal(got[i][1].view(np.uint32), ws.view(np.uint32)), (group, k, i)
    monkeypatch.delenv("VR_SPARSE_GROUP")
    # long queries: 24-32 terms each, so a group's union of terms passes the 64 the scan block holds and the host has to
    # close groups early (down to one query per group); weights with both signs (sums that cancel, negative scores)
    long_qs = []
    for i in range(40):
        m = int(rng.integers(24, 33))
        ids = rng.choice(by_df, size=m, replace=False, p=p).astype(np.int32)
        long_qs.append((ids, rng.uniform(-1.0, 1.5, size=m).astype(np.float32)))
    before = e.stats()
    got = e.search_sparse_batch(long_qs, 30)
    assert e.stats()["sparse_grouped"] - before["sparse_grouped"] == len(long_qs)
    for i, (qi, qv) in enumerate(long_qs):
        wr, ws = ocore.topk(sp.scores(qi, qv), 30, live)
        assert np.array_equal(got[i][0], wr), i
        assert np.array_equal(got[i][1].view(np.uint32), ws.view(np.uint32)), i
    want = e.search_sparse_batch(qs, 30)
    monkeypatch.setenv("VR_SPARSE_GROUPED", "0")
    before = e.stats()
    plain = e.search_sparse_batch(qs, 30)
    assert e.stats()["sparse_grouped"] == before["sparse_grouped"]
    monkeypatch.delenv("VR_SPARSE_GROUPED")
    monkeypatch.setenv("VR_SPARSE_GROUP_CAP", "2")     # two keys per (query, segment) region: the rest spills ...
    before = e.stats()
    spilled = e.search_sparse_batch(qs, 30)
    assert e.stats()["sparse_group_redo"] == before["sparse_group_redo"]
    monkeypatch.setenv("VR_SPARSE_GROUP_SPILL", "4")   # ... and a spill area of four keys overflows: those queries are redone
    before = e.stats()
    redone = e.search_sparse_batch(qs, 30)
    assert e.stats()["sparse_group_redo"] > before["sparse_group_redo"]         # (counts the queries that were redone)
    for a, b, c, d in zip(want, plain, redone, spilled):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32))
        assert np.array_equal(a[0], d[0]) and np.array_equal(a[1].view(np.uint32), d[1].view(np.uint32))


def test_sparse_batch_with_given_weights(small):
    e, rng, x, sp, folder, live, vocab = small
    qs = [(np.sort(rng.choice(vocab, size=4, replace=False)).astype(np.int32) * 7919 + 13, rng.uniform(0.1, 3.0, size=4).astype(np.float32))
          for _ in range(20)]
    got = e.search_sparse_batch(qs, 30, weights_given=True)
    for (qi, qv), (r, s) in zip(qs, got):
        r1, s1 = e.search_sparse(qi, qv, 30, weights_given=True)
        assert np.array_equal(r, r1) and np.array_equal(s.view(np.uint32), s1.view(np.uint32))


@pytest.mark.parametrize("fusion", ["minmax", "rrf"])
def test_hybrid_batch_against_the_oracle_and_the_single_search(small, fusion):
    from voitta_rag_amd import SearchFilter
    from voitta_rag_amd.engine import VR_FUSION_MINMAX, VR_FUSION_RRF

    e, rng, x, sp, folder, live, vocab = small
    nq, limit, w = 150, 10, 0.3
    mode = VR_FUSION_MINMAX if fusion == "minmax" else VR_FUSION_RRF
    q = rng.standard_normal((nq, x.shape[1])).astype(np.float32)
    sq = _queries(rng, nq, vocab)
    xh = ocore.cosine_preprocess(x)
    dsc = ocore.dense_scores(ocore.cosine_preprocess(q), xh)
    for flt, mask in ((None, live.astype(bool)), (SearchFilter(exclude_folders=[0, 2]), live.astype(bool) & ~np.isin(folder, [0, 2]))):
        before = e.stats()
        got = e.search_hybrid_batch(q, sq, limit, w, fusion=mode, flt=flt)
        assert e.stats()["batched"] - before["batched"] == nq          # the dense leg took the integer-GEMM batch
        for i in range(nq):
            dr, ds = ocore.topk(dsc[i], 3 * limit, mask.astype(np.uint8))
            if len(sq[i][0]):
                sr, ss = ocore.topk(sp.scores(*sq[i]), 3 * limit, mask.astype(np.uint8))
            else:
                sr, ss = np.zeros(0, np.int64), np.zeros(0, np.float32)
            if fusion == "minmax":
                want = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit, w, "json")
            else:
                want = ofus.rrf_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit)
            rows, scores, fd = got[i]
            assert rows.tolist() == [r for r, _, _ in want], (fusion, i)
            assert scores.tolist() == [s for _, s, _ in want], (fusion, i)
            if i % 10 == 0:
                r1, s1, f1 = e.search_hybrid(q[i], sq[i][0], sq[i][1], limit, w, fusion=mode, flt=flt)
                assert rows.tolist() == r1.tolist() and scores.tolist() == s1.tolist() and fd.tolist() == f1.tolist()


def test_hybrid_keys_merge_and_fuse_equal_the_whole(gpu):
    """What sharded.py does with the engine's own kernels: two shards (rows dealt round-robin) return their keys
    (vr_search_hybrid_keys, weights from global statistics), the merge (vr_merge_keys) yields global ids, the fusion of
    the merged lists (vr_fuse_batch) equals the single engine's hybrid answers; the df exchange (vr_sparse_row_ids /
    vr_df_apply) makes each shard's statistic the collection's."""
    from voitta_rag_amd import Engine
    from voitta_rag_amd.engine import fuse_batch

    rng = np.random.default_rng(5)
    n, dim, vocab, world = 36_000, 128, 400, 2
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x[1000:1040] = x[1000]                                            # ties across the shards
    sp = _sparse_rows(rng, n, vocab, lo=1, hi=30)
    whole = Engine(dim)
    whole.upsert(x, sparse=sp)
    shards = [Engine(dim) for _ in range(world)]
    for p, s in enumerate(shards):
        s.upsert(x[p::world], sparse=sp[p::world])
    # index-time exchange: every shard applies the OTHER shard's term ids
    exported = [s.sparse_row_ids(np.arange(n // world)) for s in shards]
    for p, s in enumerate(shards):
        for o in range(world):
            if o != p:
                s.df_apply(exported[o][0], exported[o][1], +1)
    probe = (np.arange(0, vocab, 7).astype(np.int32) * 7919 + 13)
    wdf, wn = whole.sparse_stats(probe)
    for s in shards:
        df, npts = s.sparse_stats(probe)
        assert np.array_equal(df, wdf) and npts == wn == n
    # a delete on shard 1, announced to shard 0 before it happens
    gone = np.arange(0, 400, 3)
    ids, pts = shards[1].sparse_row_ids(gone)
    shards[0].df_apply(ids, pts, -1)
    shards[1].delete_rows(gone)
    whole.delete_rows(gone * world + 1)
    wdf, wn = whole.sparse_stats(probe)
    for s in shards:
        df, npts = s.sparse_stats(probe)
        assert np.array_equal(df, wdf) and npts == wn
    nq, limit = 60, 10
    k = 3 * limit
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[0] = x[1000]
    sq = _queries(rng, nq, vocab, max_terms=6)
    parts = np.stack([s.search_hybrid_keys(q, sq, k) for s in shards])            # (world, nq, 2, k): what an all_gather yields
    gids, scores, counts = shards[0].merge_keys(parts, k)
    gids, scores, counts = gids.reshape(nq, 2, k), scores.reshape(nq, 2, k), counts.reshape(nq, 2)
    rows, fused, fd, cnt = fuse_batch(gids[:, 0], scores[:, 0], counts[:, 0], gids[:, 1], scores[:, 1], counts[:, 1], limit, 0.25)
    want = whole.search_hybrid_batch(q, sq, limit, 0.25)
    for i in range(nq):
        c = int(cnt[i])
        assert rows[i, :c].tolist() == want[i][0].tolist(), i       # global id row * world + shard == the whole's row
        assert fused[i, :c].tolist() == want[i][1].tolist() and fd[i, :c].tolist() == want[i][2].tolist()
    # the same merge with the parts in device memory (what the RCCL path hands over)
    import torch

    dev_parts = torch.from_numpy(parts.view(np.int64)).cuda()
    g2, s2, c2 = shards[1].merge_keys(dev_parts, k)
    assert np.array_equal(g2.reshape(nq, 2, k), gids) and np.array_equal(s2.reshape(nq, 2, k).view(np.uint32), scores.view(np.uint32))
    assert np.array_equal(c2.reshape(nq, 2), counts)
    for s in (whole, *shards):
        s.close()
