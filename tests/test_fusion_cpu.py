"""vr_fuse_minmax (host C++ in libvoitta_engine.so) against the Python restatement of the
reference's _hybrid_search arithmetic (oracle/fusion.py) and hand-derived known answers.
No GPU needed: the fusion entry point is host-only."""
import numpy as np
import pytest

from oracle import fusion as ofus
from voitta_rag_amd.engine import fuse_minmax


def _run(dense, sparse, limit, w, json_scores=True):
    dr = [r for r, _ in dense]
    ds = [s for _, s in dense]
    sr = [r for r, _ in sparse]
    ss = [s for _, s in sparse]
    rows, scores, fd = fuse_minmax(dr, ds, sr, ss, limit, w, json_scores)
    want = ofus.hybrid_fuse(dense, sparse, limit, w, "json" if json_scores else "exact")
    assert [int(r) for r in rows] == [r for r, _, _ in want]
    assert [float(s) for s in scores] == [s for _, s, _ in want]  # bit-exact f64
    assert [bool(f) for f in fd] == [f for _, _, f in want]
    return rows, scores, fd


def test_known_answer_overlap():
    # dense: ids 1,2,3 scores .9,.5,.1 -> norm 1, .5, 0 ; sparse: ids 3,4 scores 8,2 -> 1, 0
    rows, scores, fd = _run([(1, 0.9), (2, 0.5), (3, 0.1)], [(3, 8.0), (4, 2.0)], 10, 0.25, json_scores=True)
    assert list(rows) == [1, 2, 3, 4]
    np.testing.assert_allclose(scores, [0.75, 0.375, 0.25, 0.0], rtol=0, atol=1e-15)
    assert list(fd) == [1, 1, 1, 0]


def test_single_element_lists_normalise_to_one():
    rows, scores, _ = _run([(7, 0.3)], [(9, 5.0)], 10, 0.1)
    assert list(rows) == [7, 9]
    assert list(scores) == [0.9, 0.1]


def test_all_equal_scores_and_empty_sparse():
    rows, scores, _ = _run([(5, 0.4), (3, 0.4), (4, 0.4)], [], 2, 0.1)
    assert list(rows) == [3, 4]  # ties -> lower id (documented deviation)
    assert list(scores) == [0.9, 0.9]


@pytest.mark.parametrize("w", [0.0, 0.1, 1.0])
def test_disjoint_and_weights(w):
    _run([(1, 0.8), (2, 0.7)], [(3, 3.0), (4, 1.0)], 3, w)


@pytest.mark.parametrize("json_scores", [True, False])
def test_random_lists(json_scores):
    rng = np.random.default_rng(7)
    for trial in range(200):
        nd, ns = rng.integers(0, 31, size=2)
        ids = rng.permutation(80)
        dense = sorted(((int(i), float(np.float32(rng.uniform(-1, 1)))) for i in ids[:nd]), key=lambda t: -t[1])
        sparse = sorted(((int(i), float(np.float32(rng.uniform(0, 30)))) for i in rng.permutation(80)[:ns]),
                        key=lambda t: -t[1])
        _run(dense, sparse, int(rng.integers(1, 21)), float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])), json_scores)


# ---- reciprocal-rank fusion (north_star names it; the reference's own note on RRF: vector_store.py:638-639) ----

def _run_rrf(dense_ids, sparse_ids, limit):
    from voitta_rag_amd.engine import fuse_rrf

    rows, scores, fd = fuse_rrf(dense_ids, sparse_ids, limit)
    want = ofus.rrf_fuse([(r, 0.0) for r in dense_ids], [(r, 0.0) for r in sparse_ids], limit)
    assert [int(r) for r in rows] == [r for r, _, _ in want]
    assert [float(s) for s in scores] == [s for _, s, _ in want]  # bit-exact f64
    assert [bool(f) for f in fd] == [f for _, _, f in want]
    return rows, scores, fd


def test_rrf_known_answer_and_ties():
    # position p contributes 1 / (p + 2): id 3 is 3rd in dense (1/4) and 1st in sparse (1/2)
    rows, scores, fd = _run_rrf([1, 2, 3], [3, 4], 10)
    assert list(rows) == [3, 1, 2, 4]  # 0.75, 0.5, then the tie 1/3 == 1/3 goes to the lower row id
    assert list(scores) == [0.25 + 0.5, 0.5, 1.0 / 3.0, 1.0 / 3.0]
    assert list(fd) == [1, 1, 1, 0]
    # a full tie: the two lists are each other's mirror image
    rows, scores, _ = _run_rrf([5, 9], [9, 5], 2)
    assert list(rows) == [5, 9] and scores[0] == scores[1] == 0.5 + 1.0 / 3.0


def test_rrf_empty_sides_and_limit():
    assert len(_run_rrf([], [], 5)[0]) == 0
    assert list(_run_rrf([7, 8, 9], [], 2)[0]) == [7, 8]
    assert list(_run_rrf([], [4, 2], 5)[0]) == [4, 2]


def test_rrf_random_lists():
    rng = np.random.default_rng(11)
    for _ in range(300):
        nd, ns = rng.integers(0, 91, size=2)
        _run_rrf(rng.permutation(200)[:nd].tolist(), rng.permutation(200)[:ns].tolist(), int(rng.integers(1, 31)))


def test_json_transport_fast_path_equals_python_float_of_the_shortest_decimal():
    """fusion.cpp parses the shortest round-tripping decimal of an f32 itself when it is <= 15 digits times a power of
    ten within 10^+-22 (one correctly rounded operation) and leaves the rest to strtod. Triples of NEIGHBOURING f32
    values across 40 decades: the normalised middle value (pb - pa) / (pc - pa) moves if any of the three parses is off
    by one ulp of f64."""
    from voitta_rag_amd.engine import fuse_minmax

    rng = np.random.default_rng(1)
    for _ in range(3000):
        a = np.float32(10.0 ** rng.uniform(-30, 30) * rng.choice([-1, 1]))
        b = a
        for _ in range(int(rng.integers(1, 40))):
            b = np.nextafter(b, np.float32(np.inf), dtype=np.float32)
        c = b
        for _ in range(int(rng.integers(1, 40))):
            c = np.nextafter(c, np.float32(np.inf), dtype=np.float32)
        pa, pb, pc = (float(str(v)) for v in (a, b, c))
        rows, scores, _ = fuse_minmax([0, 1, 2], [a, b, c], [], [], 3, 0.0, True)
        assert dict(zip(rows.tolist(), scores.tolist()))[1] == (pb - pa) / (pc - pa), (a, b, c)


def test_fuse_batch_equals_fuse_per_query_and_survives_repeated_and_concurrent_calls():
    """vr_fuse_batch runs on the library's parked host threads (csrc/host_parallel.h): many calls in a row, and from
    several Python threads at once (the loop that finds the pool busy runs inline), return what vr_fuse_minmax /
    vr_fuse_rrf return per query."""
    import threading

    from voitta_rag_amd.engine import VR_FUSION_RRF, fuse_batch, fuse_minmax, fuse_rrf

    rng = np.random.default_rng(4)
    nq, k, limit = 300, 30, 10
    d_ids = np.stack([rng.choice(5000, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
    s_ids = np.stack([rng.choice(5000, size=k, replace=False) for _ in range(nq)]).astype(np.int64)
    d_sc = -np.sort(-rng.random((nq, k)).astype(np.float32), axis=1)
    s_sc = -np.sort(-(rng.random((nq, k)) * 20).astype(np.float32), axis=1)
    d_cnt = rng.integers(0, k + 1, size=nq).astype(np.int32)
    s_cnt = rng.integers(0, k + 1, size=nq).astype(np.int32)
    want = [fuse_minmax(d_ids[i, :d_cnt[i]], d_sc[i, :d_cnt[i]], s_ids[i, :s_cnt[i]], s_sc[i, :s_cnt[i]], limit, 0.3, True) for i in range(nq)]
    want_rrf = [fuse_rrf(d_ids[i, :d_cnt[i]], s_ids[i, :s_cnt[i]], limit) for i in range(nq)]

    def check():
        rows, scores, fd, cnt = fuse_batch(d_ids, d_sc, d_cnt, s_ids, s_sc, s_cnt, limit, 0.3)
        for i in range(nq):
            c = int(cnt[i])
            assert rows[i, :c].tolist() == want[i][0].tolist() and scores[i, :c].tolist() == want[i][1].tolist()
            assert fd[i, :c].tolist() == want[i][2].tolist()
        rows, scores, fd, cnt = fuse_batch(d_ids, d_sc, d_cnt, s_ids, s_sc, s_cnt, limit, 0.3, VR_FUSION_RRF)
        for i in range(0, nq, 7):
            c = int(cnt[i])
            assert rows[i, :c].tolist() == want_rrf[i][0].tolist() and scores[i, :c].tolist() == want_rrf[i][1].tolist()

    for _ in range(5):
        check()
    errors = []

    def worker():
        try:
            for _ in range(3):
                check()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors
