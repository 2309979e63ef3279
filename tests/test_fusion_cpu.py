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
