"""BASELINE configs[2] size (1M x 768) — where the CPU oracle is too slow to be the checker, parity
is held through size-independent properties (the contract's list: sortedness, idempotence, merge of
parts = whole, two independent code paths agreeing bit for bit) plus a spot check of the winners
against the oracle's own arithmetic on the rows that matter."""
import numpy as np
import pytest

from oracle import core as ocore

pytestmark = pytest.mark.gpu

N, D = 1_000_000, 768


@pytest.fixture(scope="module")
def corpus(gpu):
    import torch

    from voitta_rag_amd import Engine

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(2026)
    two = Engine(D, initial_rows=N)                      # f16 shadow + two-stage search
    one = Engine(D, initial_rows=N, prefilter=False)     # one-stage f32 scan only
    halves = [Engine(D, initial_rows=N // 2, prefilter=False), Engine(D, initial_rows=N // 2)]
    nnz = 24
    keep = []
    for a in range(0, N, 125_000):
        x = torch.randn((125_000, D), device=dev, generator=g)
        x[:1000] = x[0] + 0.05 * torch.randn((1000, D), device=dev, generator=g)   # a cluster per block
        ids = (torch.rand((125_000, nnz), device=dev, generator=g) ** 2 * 50_000).to(torch.int32)
        ids, _ = torch.sort(ids, dim=1)
        ids = ids * 32 + torch.arange(nnz, device=dev, dtype=torch.int32)[None, :]   # distinct, ascending
        off = (torch.arange(125_001, device=dev, dtype=torch.int64) * nnz).contiguous()
        val = torch.rand((125_000 * nnz,), device=dev, generator=g) + 0.5
        sp = (off, ids.reshape(-1).contiguous(), val.contiguous())
        for e in (two, one, halves[0] if a < N // 2 else halves[1]):
            e.upsert(x.contiguous(), sparse=sp)
        keep.append(x[:2000].cpu().numpy())              # enough to spot-check winners from the clusters
    q = torch.randn((24, D), device=dev, generator=g)
    q[:8] = torch.cat([torch.from_numpy(k[:1]) for k in keep]).to(dev) + 0.01 * torch.randn((8, D), device=dev, generator=g)
    yield two, one, halves, q.cpu().numpy(), keep
    for e in (two, one, *halves):
        e.close()


def test_two_code_paths_agree_bit_for_bit_and_results_are_sorted(corpus):
    two, one, halves, q, keep = corpus
    for i in range(q.shape[0]):
        for k in (10, 30, 64):
            r2, s2 = two.search_dense(q[i:i + 1], k)[0]
            r1, s1 = one.search_dense(q[i:i + 1], k)[0]
            assert np.array_equal(r2, r1) and np.array_equal(s2.view(np.uint32), s1.view(np.uint32)), (i, k)
            assert len(r2) == k and len(set(r2.tolist())) == k
            assert np.all(s2[:-1] >= s2[1:])                                             # sorted
            ties = s2[:-1] == s2[1:]
            assert np.all(r2[:-1][ties] < r2[1:][ties])                                  # ties: lower row first
            ra, sa = two.search_dense(q[i:i + 1], k)[0]                                  # idempotent
            assert np.array_equal(ra, r2) and np.array_equal(sa, s2)
    st = two.stats()
    assert st["two_stage"] >= q.shape[0] * 6 and st["fallback"] == 0
    assert one.stats()["two_stage"] == 0
    # a 16-query block through the batched one-stage scan equals the single-query answers
    block = two.search_dense(q[:16], 10)
    for i in range(16):
        r, s = two.search_dense(q[i:i + 1], 10)[0]
        assert np.array_equal(block[i][0], r) and np.array_equal(block[i][1], s)


def test_merge_of_halves_equals_whole(corpus):
    """Sharding invariance at full size: top-k of the union = merge of the per-shard top-k lists
    (what sharded.py does over RCCL), dense and sparse, with global document frequencies."""
    two, one, halves, q, keep = corpus
    for i in range(0, q.shape[0], 3):
        want_r, want_s = two.search_dense(q[i:i + 1], 30)[0]
        parts = [h.search_dense(q[i:i + 1], 30)[0] for h in halves]
        rows = np.concatenate([parts[0][0], parts[1][0] + N // 2])
        scores = np.concatenate([parts[0][1], parts[1][1]])
        order = np.lexsort((rows, -scores.astype(np.float64)))[:30]
        assert np.array_equal(rows[order], want_r) and np.array_equal(scores[order], want_s)
    qi = np.asarray([3 * 32 + 0, 100 * 32 + 1, 2000 * 32 + 5, 40_000 * 32 + 20], np.int32)
    df = [sum(int(h.sparse_stats(qi)[0][j]) for h in halves) for j in range(len(qi))]
    whole_df, n_pts = two.sparse_stats(qi)
    assert [int(v) for v in whole_df] == df and n_pts == N
    w = np.asarray([two.idf(N, d) for d in df], np.float32)                              # global statistics
    want_r, want_s = two.search_sparse(qi, np.ones(4, np.float32), 30)
    parts = [h.search_sparse(qi, w, 30, weights_given=True) for h in halves]
    rows = np.concatenate([parts[0][0], parts[1][0] + N // 2])
    scores = np.concatenate([parts[0][1], parts[1][1]])
    order = np.lexsort((rows, -scores.astype(np.float64)))[:30]
    assert np.array_equal(rows[order], want_r) and np.array_equal(scores[order], want_s)


def test_winners_match_the_oracle_arithmetic(corpus):
    """The first 8 queries sit next to the cluster heads, so their winners are among the rows kept
    on the host: the engine's f32 scores for those rows must equal the oracle's fma chain bit for bit,
    and every kept row must rank exactly where its oracle score puts it."""
    two, one, halves, q, keep = corpus
    for b in range(8):
        rows, scores = two.search_dense(q[b:b + 1], 64)[0]
        base = b * 125_000
        local = (rows >= base) & (rows < base + 2000)
        assert local.sum() >= 32                                                         # the cluster dominates
        xh = ocore.cosine_preprocess(keep[b])
        sc = ocore.dense_scores(ocore.cosine_preprocess(q[b:b + 1]), xh)[0]
        got = scores[local]
        want = sc[rows[local] - base]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # no kept row outside the result may beat the last returned score
        outside = np.setdiff1d(np.arange(2000), rows[local] - base)
        assert np.all(sc[outside] <= scores[-1])


def test_deletes_at_full_size(corpus):
    two, one, halves, q, keep = corpus
    r, s = two.search_dense(q[:1], 10)[0]
    for e in (two, one):
        e.delete_rows(r[:5])
    r2, s2 = two.search_dense(q[:1], 10)[0]
    r1, s1 = one.search_dense(q[:1], 10)[0]
    assert np.array_equal(r2, r1) and np.array_equal(s2, s1)
    assert not set(r[:5].tolist()) & set(r2.tolist()) and np.array_equal(r2[:5], r[5:])
    assert two.count() == (N, N - 5)


def test_inverted_and_forward_sparse_scans_agree_at_full_size(corpus, monkeypatch):
    """Two independent code paths for the sparse leg — the postings of the query's terms (csrc/invert.hip, what runs
    by default) and the scan of every stored id (csrc/sparse.hip) — must return the same rows and the same f32 bits
    over 1M rows x 24 terms, alone and inside a hybrid search (both fusion modes), before and after deletes."""
    from voitta_rag_amd.engine import VR_FUSION_RRF

    two, one, halves, q, keep = corpus
    rng = np.random.default_rng(31)

    def both(fn):
        monkeypatch.setenv("VR_SPARSE_INVERTED", "1")
        a = fn()
        monkeypatch.setenv("VR_SPARSE_INVERTED", "0")
        b = fn()
        monkeypatch.setenv("VR_SPARSE_INVERTED", "1")
        return a, b

    def check():
        for trial in range(12):
            m = int(rng.integers(1, 9)) if trial < 10 else 32
            slot = rng.choice(24, size=min(m, 24), replace=False)
            base = (rng.random(len(slot)) ** 2 * 50_000).astype(np.int64)            # the corpus's own id distribution
            if m == 32:                                                              # the most terms the postings serve
                slot = np.concatenate([slot, rng.choice(24, size=8)])
                base = np.concatenate([base, rng.integers(0, 50_000, size=8)])
            qi = np.unique((base * 32 + slot).astype(np.int32))
            qv = rng.uniform(0.5, 1.5, size=len(qi)).astype(np.float32)
            for k in (10, 30, 64):
                (ri, si), (rf, sf) = both(lambda: two.search_sparse(qi, qv, k))
                assert np.array_equal(ri, rf) and np.array_equal(si.view(np.uint32), sf.view(np.uint32)), (trial, k)
                assert np.all(si[:-1] >= si[1:])
            for fusion in (None, VR_FUSION_RRF):
                kw = {} if fusion is None else {"fusion": fusion}
                a, b = both(lambda: two.search_hybrid(q[trial], qi, qv, 10, 0.3, **kw))
                assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() and a[2].tolist() == b[2].tolist()

    check()
    r, _ = two.search_sparse(np.asarray([3 * 32 + 0, 100 * 32 + 1], np.int32), np.ones(2, np.float32), 30)
    two.delete_rows(r[:10])
    one.delete_rows(r[:10])   # (the two engines of the fixture stay the same collection)
    check()


def test_batched_queries_equal_single_queries_at_full_size(corpus):
    """configs[4]'s query side on one shard: 1000 queries in ONE call (integer-GEMM batched search) must return, for
    every query, exactly what 1000 single searches return (two-stage single-query path), and both must agree with the
    one-stage f32 engine on a sample."""
    two, one, halves, q, keep = corpus
    rng = np.random.default_rng(4)
    qs = rng.standard_normal((1000, D)).astype(np.float32)
    qs[:24] = q
    before = two.stats()
    batched = two.search_dense(qs, 10)
    after = two.stats()
    assert after["batched"] - before["batched"] == 1000
    for i in range(0, 1000, 7):
        r1, s1 = two.search_dense(qs[i:i + 1], 10)[0]
        assert np.array_equal(batched[i][0], r1) and np.array_equal(batched[i][1].view(np.uint32), s1.view(np.uint32)), i
    for i in range(0, 24):
        r1, s1 = one.search_dense(qs[i:i + 1], 10)[0]
        assert np.array_equal(batched[i][0], r1) and np.array_equal(batched[i][1].view(np.uint32), s1.view(np.uint32)), i
    # 16 queries at a time through the exact f32 scan (the pre-batch path) for a block of them
    blk = one.search_dense(qs[100:116], 10)
    for j in range(16):
        assert np.array_equal(batched[100 + j][0], blk[j][0]) and np.array_equal(batched[100 + j][1], blk[j][1])


def test_configs4_shard_batched_equals_single_queries_1p25m_x_1024(gpu):
    """BASELINE configs[4], one rank's share: a 1.25M x 1024 (bge-large width) shard of the 10M corpus, 1000 queries in
    ONE call. The batched answers must equal 1000 single two-stage searches bit for bit — every one of them — and the
    one-stage f32 scan on a sample; results sorted, ties by row. Rows share a common direction (real collections do),
    so the centred shadow is what keeps the batch inside its candidate budget."""
    import torch

    from voitta_rag_amd import Engine

    n, d = 1_250_000, 1024
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(44)
    common = torch.nn.functional.normalize(torch.randn(d, device=dev, generator=g), dim=0)
    two = Engine(d, initial_rows=n)
    one = Engine(d, initial_rows=n, prefilter=False)
    for a in range(0, n, 125_000):
        x = torch.randn((125_000, d), device=dev, generator=g)
        x = torch.nn.functional.normalize(x, dim=1) * (0.5 ** 0.5) + (0.5 ** 0.5) * common[None, :]   # pairwise cos ~0.5
        x[:500] = x[0] + 0.02 * torch.randn((500, d), device=dev, generator=g)                          # a cluster per block
        two.upsert(x.contiguous())
        one.upsert(x.contiguous())
    qs = torch.nn.functional.normalize(torch.randn((1000, d), device=dev, generator=g), dim=1) * (0.5 ** 0.5) + (0.5 ** 0.5) * common[None, :]
    qs = qs.cpu().numpy()
    before = two.stats()
    batched = two.search_dense(qs, 10)
    after = two.stats()
    assert after["batched"] - before["batched"] == 1000
    assert after["batch_fallback"] - before["batch_fallback"] < 50, after  # (a fallback is still exact; it must stay rare)
    for i in range(1000):
        r1, s1 = two.search_dense(qs[i:i + 1], 10)[0]
        assert np.array_equal(batched[i][0], r1) and np.array_equal(batched[i][1].view(np.uint32), s1.view(np.uint32)), i
        assert np.all(s1[:-1] >= s1[1:]) and np.all(r1[:-1][s1[:-1] == s1[1:]] < r1[1:][s1[:-1] == s1[1:]])
    assert two.stats()["fallback"] == before["fallback"]
    for i in range(0, 1000, 50):
        r1, s1 = one.search_dense(qs[i:i + 1], 10)[0]
        assert np.array_equal(batched[i][0], r1) and np.array_equal(batched[i][1].view(np.uint32), s1.view(np.uint32)), i
    two.close()
    one.close()


def test_configs3_shard_dense_only_1p25m_x_384(gpu):
    """BASELINE configs[3], one rank's share: a 1.25M x 384 (MiniLM width) dense-only shard of the 10M corpus. The
    two-stage search (int8 shadow) must equal the one-stage f32 scan bit for bit, the merge of two half shards must
    equal the whole (what the RCCL top-k merge does), the winners' scores must be the oracle's fma chain on the rows
    kept on the host, and a 1000-query batch must equal the single searches (reference: vector_store.py:612-617)."""
    import torch

    from voitta_rag_amd import Engine

    n, d, blk = 1_250_000, 384, 125_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(33)
    two = Engine(d, initial_rows=n)
    one = Engine(d, initial_rows=n, prefilter=False)
    halves = [Engine(d, initial_rows=n // 2), Engine(d, initial_rows=n // 2, prefilter=False)]
    keep = []
    for a in range(0, n, blk):
        x = torch.randn((blk, d), device=dev, generator=g)
        x[:1000] = x[0] + 0.05 * torch.randn((1000, d), device=dev, generator=g)       # a cluster per block
        for e in (two, one, halves[0] if a < n // 2 else halves[1]):
            e.upsert(x.contiguous())
        keep.append(x[:2000].cpu().numpy())
    q = torch.randn((20, d), device=dev, generator=g)
    q[:10] = torch.cat([torch.from_numpy(k[:1]) for k in keep]).to(dev) + 0.01 * torch.randn((10, d), device=dev, generator=g)
    q = q.cpu().numpy()
    assert two.count() == (n, n) and two.sparse_stats(np.asarray([1], np.int32))[1] == 0   # dense-only: no sparse points
    for i in range(q.shape[0]):
        for k in (10, 30):
            r2, s2 = two.search_dense(q[i:i + 1], k)[0]
            r1, s1 = one.search_dense(q[i:i + 1], k)[0]
            assert np.array_equal(r2, r1) and np.array_equal(s2.view(np.uint32), s1.view(np.uint32)), (i, k)
            assert len(set(r2.tolist())) == k and np.all(s2[:-1] >= s2[1:])
            ties = s2[:-1] == s2[1:]
            assert np.all(r2[:-1][ties] < r2[1:][ties])
        # merge of the half shards = the whole
        want_r, want_s = two.search_dense(q[i:i + 1], 30)[0]
        parts = [h.search_dense(q[i:i + 1], 30)[0] for h in halves]
        rows = np.concatenate([parts[0][0], parts[1][0] + n // 2])
        scores = np.concatenate([parts[0][1], parts[1][1]])
        order = np.lexsort((rows, -scores.astype(np.float64)))[:30]
        assert np.array_equal(rows[order], want_r) and np.array_equal(scores[order], want_s)
    st = two.stats()
    assert st["two_stage"] >= q.shape[0] * 3 and st["fallback"] == 0
    for b in range(10):  # winners against the oracle's arithmetic
        rows, scores = two.search_dense(q[b:b + 1], 64)[0]
        base = b * blk
        local = (rows >= base) & (rows < base + 2000)
        assert local.sum() >= 32
        sc = ocore.dense_scores(ocore.cosine_preprocess(q[b:b + 1]), ocore.cosine_preprocess(keep[b]))[0]
        assert np.array_equal(scores[local].view(np.uint32), sc[rows[local] - base].view(np.uint32))
        outside = np.setdiff1d(np.arange(2000), rows[local] - base)
        assert np.all(sc[outside] <= scores[-1])
    # the batched form at this width (D = 384 = three 128-deep K-tiles of the integer GEMM)
    rng = np.random.default_rng(6)
    qs = rng.standard_normal((1000, d)).astype(np.float32)
    qs[:20] = q
    before = two.stats()
    batched = two.search_dense(qs, 10)
    assert two.stats()["batched"] - before["batched"] == 1000
    for i in list(range(0, 1000, 9)) + list(range(20)):
        r1, s1 = two.search_dense(qs[i:i + 1], 10)[0]
        assert np.array_equal(batched[i][0], r1) and np.array_equal(batched[i][1].view(np.uint32), s1.view(np.uint32)), i
    for i in range(0, 20):
        r1, s1 = one.search_dense(qs[i:i + 1], 10)[0]
        assert np.array_equal(batched[i][0], r1) and np.array_equal(batched[i][1].view(np.uint32), s1.view(np.uint32)), i
    for e in (two, one, *halves):
        e.close()


def test_configs4_shard_hybrid_batch_1p25m_x_1024_with_sparse_rows(gpu):
    """BASELINE configs[4], one rank's share WITH its sparse side: 1.25M x 1024 rows (bge-large width), 24 BM25 terms
    per row, 1000 hybrid top-10 queries in ONE call (vr_search_hybrid_batch). Every batched answer must equal the
    single-query vr_search_hybrid bit for bit (rows, f64 fused scores, origin flags) in both fusion modes, and a sample
    must equal the CPU oracle end to end: oracle dense scores (f32 fma chain) + oracle IDF sparse scores over all
    1.25M rows -> top-30 each -> oracle/fusion.py (vector_store.py:621-697)."""
    import torch

    from oracle import fusion as ofus
    from voitta_rag_amd import Engine
    from voitta_rag_amd.engine import VR_FUSION_RRF

    n, d, blk, nnz = 1_250_000, 1024, 125_000, 24
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(404)
    common = torch.nn.functional.normalize(torch.randn(d, device=dev, generator=g), dim=0)
    e = Engine(d, initial_rows=n)
    xs, idx_l, val_l = [], [], []
    for a in range(0, n, blk):
        x = torch.nn.functional.normalize(torch.randn((blk, d), device=dev, generator=g), dim=1) * (0.5 ** 0.5) + (0.5 ** 0.5) * common[None, :]
        ids = (torch.rand((blk, nnz), device=dev, generator=g) ** 2 * 50_000).to(torch.int32)
        ids, _ = torch.sort(ids, dim=1)
        ids = ids * 32 + torch.arange(nnz, device=dev, dtype=torch.int32)[None, :]          # distinct, ascending
        off = (torch.arange(blk + 1, device=dev, dtype=torch.int64) * nnz).contiguous()
        val = (torch.rand((blk * nnz,), device=dev, generator=g) + 0.5).contiguous()
        e.upsert(x.contiguous(), sparse=(off, ids.reshape(-1).contiguous(), val))
        xs.append(x.cpu().numpy())
        idx_l.append(ids.reshape(-1).cpu().numpy())
        val_l.append(val.cpu().numpy())
    nq, limit, w = 1000, 10, 0.2
    rng = np.random.default_rng(9)
    qs = (torch.nn.functional.normalize(torch.randn((nq, d), device=dev, generator=g), dim=1) * (0.5 ** 0.5) + (0.5 ** 0.5) * common[None, :]).cpu().numpy()
    sq = []
    for i in range(nq):
        m = int(rng.integers(1, 8)) if i % 50 else 0
        slot = rng.choice(nnz, size=m, replace=False)
        base = (rng.random(m) ** 2 * 50_000).astype(np.int64)
        sq.append(((base * 32 + slot).astype(np.int32), rng.uniform(0.5, 1.5, size=m).astype(np.float32)))
    gone = np.arange(5, n, 997)
    e.delete_rows(gone)
    before = e.stats()
    got = e.search_hybrid_batch(qs, sq, limit, w)
    after = e.stats()
    assert after["batched"] - before["batched"] == nq
    # the sparse legs took the grouped scan (csrc/invert.hip: groups of queries per segment block, thresholds from a
    # sample of the segments) and no batch had to be redone on the per-query kernels
    assert after["sparse_grouped"] - before["sparse_grouped"] == nq and after["sparse_group_redo"] == before["sparse_group_redo"]
    got_rrf = e.search_hybrid_batch(qs, sq, limit, w, fusion=VR_FUSION_RRF)
    for i in range(nq):
        r1, s1, f1 = e.search_hybrid(qs[i], sq[i][0], sq[i][1], limit, w)
        assert got[i][0].tolist() == r1.tolist() and got[i][1].tolist() == s1.tolist() and got[i][2].tolist() == f1.tolist(), i
        assert np.all(np.diff(got[i][1]) <= 0)
        if i % 10 == 0:
            r2, s2, f2 = e.search_hybrid(qs[i], sq[i][0], sq[i][1], limit, w, fusion=VR_FUSION_RRF)
            assert got_rrf[i][0].tolist() == r2.tolist() and got_rrf[i][1].tolist() == s2.tolist(), i
    # the sparse legs alone, batched == single
    sb = e.search_sparse_batch(sq[:200], 30)
    for i in range(0, 200, 5):
        r1, s1 = e.search_sparse(sq[i][0], sq[i][1], 30) if len(sq[i][0]) else (np.zeros(0, np.int64), np.zeros(0, np.float32))
        assert np.array_equal(sb[i][0], r1) and np.array_equal(sb[i][1].view(np.uint32), s1.view(np.uint32)), i
    # a sample against the oracle, end to end
    live = np.ones(n, np.uint8)
    live[gone] = 0
    x_all = np.concatenate(xs)
    del xs
    xh = ocore.cosine_preprocess(x_all)
    del x_all
    off_all = np.arange(n + 1, dtype=np.int64) * nnz
    so = ocore.SparseOracle.from_csr(off_all, np.concatenate(idx_l), np.concatenate(val_l), live.astype(bool))
    assert so.n_points == n - len(gone)
    for i in (0, 1, 50, 333, 999):
        dsc = ocore.dense_scores(ocore.cosine_preprocess(qs[i:i + 1]), xh)[0]
        dr, ds = ocore.topk(dsc, 3 * limit, live)
        if len(sq[i][0]):
            sr, ss = ocore.topk(so.scores(*sq[i]), 3 * limit, live)
        else:
            sr, ss = np.zeros(0, np.int64), np.zeros(0, np.float32)
        want = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit, w, "json")
        assert got[i][0].tolist() == [r for r, _, _ in want], i
        assert got[i][1].tolist() == [s for _, s, _ in want], i
    e.close()
