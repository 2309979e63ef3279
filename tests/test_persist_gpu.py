"""vr_save / vr_load: the index must survive a restart the way the reference's
does in Qdrant's volume (reference: docker-compose.yml:8-9, vector_store.py:75-115). Round trip =
identical rows, identical f32 score bits, identical counts — with tombstones, filters, sparse
vectors and further upserts after the load; corrupt or mismatching files are refused."""
import os

import numpy as np
import pytest

from oracle import core as ocore

pytestmark = pytest.mark.gpu


def _fill(e, rng, n, dim, first_folder=0):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    nnz = rng.integers(1, 30, size=n)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(nnz)
    idx = np.concatenate([np.sort(rng.choice(5000, size=k, replace=False)) for k in nnz]).astype(np.int32)
    val = (rng.random(off[-1]) + 0.25).astype(np.float32)
    folder = (rng.integers(0, 4, size=n) + first_folder).astype(np.int32)
    modified = rng.integers(1_600_000_000, 1_700_000_000, size=n).astype(np.int64)
    e.upsert(x, sparse=(off, idx, val), folder_ids=folder, modified=modified)
    return x, (off, idx, val), folder, modified


def _queries(e, rng, dim, flt):
    out = []
    for _ in range(6):
        q = rng.standard_normal(dim).astype(np.float32)
        qi = np.sort(rng.choice(5000, size=5, replace=False)).astype(np.int32)
        qv = np.ones(5, np.float32)
        out.append((e.search_dense(q[None], 10, flt)[0], e.search_dense(np.tile(q, (3, 1)), 100)[1],
                    e.search_sparse(qi, qv, 30, flt), e.search_hybrid(q, qi, qv, 10, 0.1, flt=flt)))
    return out


def _same(a, b):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y)
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    else:
        assert a == b


@pytest.mark.parametrize("dim,n", [(64, 5000), (768, 9000), (48, 700)])  # 48: no f16 shadow
def test_save_load_round_trip_is_bit_identical(gpu, tmp_path, dim, n):
    from voitta_rag_amd import Engine, SearchFilter

    rng = np.random.default_rng(dim + n)
    e = Engine(dim)
    _fill(e, rng, n, dim)
    e.delete_rows(rng.choice(n, size=n // 7, replace=False))
    _fill(e, rng, 300, dim, first_folder=2)   # a second batch: more slices, df growth
    flt = SearchFilter(include_folders=[1, 2, 4])
    path = str(tmp_path / "idx.vrindex")
    e.save(path)
    assert not os.path.exists(path + ".tmp")
    before = _queries(e, np.random.default_rng(7), dim, flt)
    counts = e.count()
    e2 = Engine(dim)
    e2.load(path)
    assert e2.count() == counts
    _same(before, _queries(e2, np.random.default_rng(7), dim, flt))
    # both engines keep working identically after the load: upsert + delete + search again
    for eng in (e, e2):
        _fill(eng, np.random.default_rng(99), 500, dim)
        eng.delete_rows(np.arange(10, 400, 7))
    _same(_queries(e, np.random.default_rng(8), dim, flt), _queries(e2, np.random.default_rng(8), dim, flt))
    assert e.count() == e2.count()
    e.close()
    e2.close()


def test_load_refuses_bad_files(gpu, tmp_path):
    from voitta_rag_amd import Engine
    from voitta_rag_amd._lib import EngineError

    rng = np.random.default_rng(3)
    e = Engine(64)
    _fill(e, rng, 1000, 64)
    path = str(tmp_path / "a.vrindex")
    e.save(path)
    with pytest.raises(EngineError, match="empty engine"):
        e.load(path)                                   # not empty
    e.close()
    with pytest.raises(EngineError, match="dimensional"):
        Engine(128).load(path)                         # wrong dimension
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0x40
    bad = str(tmp_path / "b.vrindex")
    open(bad, "wb").write(bytes(blob))
    with pytest.raises(EngineError, match="corrupt"):
        Engine(64).load(bad)                           # flipped bit
    open(bad, "wb").write(bytes(blob[: len(blob) // 3]))
    with pytest.raises(EngineError, match="truncated"):
        Engine(64).load(bad)
    open(bad, "wb").write(b"not an index at all" * 10)
    with pytest.raises(EngineError, match="not an index file"):
        Engine(64).load(bad)
    with pytest.raises(EngineError, match="cannot open"):
        Engine(64).load(str(tmp_path / "missing.vrindex"))


def test_empty_engine_round_trip(gpu, tmp_path):
    from voitta_rag_amd import Engine

    e = Engine(64)
    p = str(tmp_path / "empty.vrindex")
    e.save(p)
    e2 = Engine(64)
    e2.load(p)
    assert e2.count() == (0, 0)
    rows, scores = e2.search_dense(np.ones((1, 64), np.float32), 5)[0]
    assert len(rows) == 0
    x = np.random.default_rng(0).standard_normal((100, 64)).astype(np.float32)
    e2.upsert(x)
    sc = ocore.dense_scores(ocore.cosine_preprocess(x[:1]), ocore.cosine_preprocess(x))[0]
    wr, ws = ocore.topk(sc, 5)
    gr, gs = e2.search_dense(x[:1], 5)[0]
    assert np.array_equal(gr, wr) and np.array_equal(gs, ws)


@pytest.mark.parametrize("dim,n", [(64, 6000), (768, 5000), (48, 900)])
def test_compact_drops_tombstones_and_keeps_every_answer(gpu, dim, n):
    """vr_compact: after compaction every search returns the same documents with the
    same f32 score bits; rows are renumbered in order; the engine keeps accepting upserts/deletes and
    stays equal to an engine that was built from the surviving rows only."""
    from voitta_rag_amd import Engine, SearchFilter

    rng = np.random.default_rng(n)
    e = Engine(dim)
    x, (off, idx, val), folder, modified = _fill(e, rng, n, dim)
    x2 = rng.standard_normal((200, dim)).astype(np.float32)
    e.upsert(x2, folder_ids=np.full(200, 1, np.int32))            # rows without a sparse vector
    dead = np.sort(rng.choice(n + 200, size=(n + 200) // 3, replace=False))
    e.delete_rows(dead)
    flt = SearchFilter(include_folders=[0, 1])
    before = _queries(e, np.random.default_rng(5), dim, flt)
    n_rows, n_live = e.count()
    remap = e.compact()
    assert remap.shape == (n_rows,) and (remap[dead] == -1).all()
    keep = np.flatnonzero(remap >= 0)
    assert np.array_equal(remap[keep], np.arange(keep.size)) and e.count() == (n_live, n_live)
    after = _queries(e, np.random.default_rng(5), dim, flt)

    def renumber(res):   # map the pre-compaction rows of a result through remap, keep scores
        if isinstance(res, tuple) and len(res) and isinstance(res[0], np.ndarray) and res[0].dtype == np.int64:
            return (remap[res[0]],) + tuple(res[1:])
        if isinstance(res, (tuple, list)):
            return type(res)(renumber(r) for r in res)
        return res

    _same(renumber(before), after)
    # a second engine built from the survivors only must be indistinguishable, also after more traffic
    e2 = Engine(dim)
    allx = np.concatenate([x, x2])
    sp_rows = [(idx[off[r]:off[r + 1]], val[off[r]:off[r + 1]]) for r in range(n)]
    k0 = keep[keep < n]
    e2.upsert(allx[k0], sparse=[sp_rows[r] for r in k0], folder_ids=folder[k0], modified=modified[k0])
    k1 = keep[keep >= n]
    e2.upsert(allx[k1], folder_ids=np.full(k1.size, 1, np.int32))
    _same(after, _queries(e2, np.random.default_rng(5), dim, flt))
    for eng in (e, e2):
        _fill(eng, np.random.default_rng(42), 300, dim)
        eng.delete_rows(np.arange(5, 900, 11))
    _same(_queries(e, np.random.default_rng(6), dim, flt), _queries(e2, np.random.default_rng(6), dim, flt))
    assert e.count() == e2.count()
    assert e.compact().shape[0] == e2.count()[0] and e.count()[0] == e.count()[1]
    e.close()
    e2.close()
