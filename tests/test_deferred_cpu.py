"""Write-behind of the unmodified caller sequence (voitta_rag_amd/deferred.py), host logic only:
the engine here is the oracle double (tests/oracle_engine.py) extended by an ``index_batch`` whose "encoder" is a
fixed function of the token ids, so that the test can tell WHICH tokens became WHICH row. What is checked is the
plumbing: row order, read-your-writes, lazy materialisation, subsets, failure handling. The numerics of the real fused
call are the GPU tier's business (tests/test_services_gpu.py)."""
import threading
import time
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import core as ocore
from oracle_engine import OracleEngine

DIM = 32


def toy_embed(ids, off):
    """One deterministic vector per token sequence."""
    out = np.zeros((len(off) - 1, DIM), np.float32)
    for i in range(len(off) - 1):
        seq = np.asarray(ids[off[i]:off[i + 1]], np.int64)
        for j, t in enumerate(seq):
            out[i, (t * 7 + j) % DIM] += 1.0 + (t % 5)
    return out


def toy_tf(off, stems):
    rows = []
    for i in range(len(off) - 1):
        ids, cnt = np.unique(np.asarray(stems[off[i]:off[i + 1]], np.int32), return_counts=True)
        rows.append((ids.astype(np.int32), (cnt / (cnt + 1.0)).astype(np.float32)))
    return rows


class ToyEngine(OracleEngine):
    def __init__(self, dim):
        super().__init__(dim)
        self.batches = []       # chunk count of every fused call
        self.fail_next = False
        self.encodes = 0
        self.delay = 0.0

    def index_batch(self, wp_ids, wp_off, bm_ids=None, bm_off=None, folder_ids=None, index_folder_ids=None,
                    created=None, modified=None):
        if self.fail_next:
            self.fail_next = False
            raise RuntimeError("engine said no")
        time.sleep(self.delay)
        n = len(wp_off) - 1
        self.batches.append(n)
        sparse = toy_tf(bm_off, bm_ids) if bm_ids is not None else None
        return self.upsert(toy_embed(wp_ids, wp_off), sparse=sparse, folder_ids=folder_ids, index_folder_ids=index_folder_ids,
                           created=created, modified=modified)

    def bm25_tf(self, off, stems):
        return toy_tf(off, stems)


class ToyEncoder:
    def __init__(self, engine):
        self.engine = engine
        self.desc = SimpleNamespace(hidden=DIM)


@pytest.fixture
def store(monkeypatch):
    from voitta_rag_amd import config, deferred, store_registry, vector_store
    from voitta_rag_amd import encoder as enc

    monkeypatch.setenv("EMBEDDING_DIMENSION", str(DIM))
    monkeypatch.setenv("VOITTA_DEFERRED_INDEXING", "1")  # the opt-in write-behind; the default is checked by its own test below
    config.get_settings.cache_clear()
    engine = ToyEngine(DIM)
    store_registry.set_engine(engine)

    def fake_encode(eng, ids, off):
        eng.encodes += 1
        return toy_embed(ids, off)

    monkeypatch.setattr(enc, "encode", fake_encode)
    vs = vector_store.VectorStoreService()
    encoder = ToyEncoder(engine)
    rng = np.random.default_rng(0)

    def file_of(name, n):
        """What embed_texts / sparse embed_texts / the chunker hand to store_chunks for one file."""
        lens = rng.integers(3, 9, size=n)
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum(lens)
        ids = rng.integers(5, 400, size=int(off[-1])).astype(np.int32)
        slens = rng.integers(1, 6, size=n)
        soff = np.zeros(n + 1, np.int64)
        soff[1:] = np.cumsum(slens)
        stems = rng.integers(1, 60, size=int(soff[-1])).astype(np.int32)
        emb = deferred.DeferredEmbeddings(encoder, ids, off)
        sp = deferred.DeferredSparse(engine, soff, stems)
        metas = [vector_store.ChunkMetadata(file_path=name, folder_path="d", index_folder="d", file_name=name, chunk_index=i,
                                            total_chunks=n, start_char=0, end_char=1, indexed_at="t") for i in range(n)]
        return emb, sp, metas, (ids, off, soff, stems)

    yield SimpleNamespace(vs=vs, engine=engine, file_of=file_of, deferred=deferred)
    store_registry.set_engine(None)
    config.get_settings.cache_clear()


def test_untouched_embeddings_are_stored_by_fused_calls_in_row_order(store):
    vs, engine = store.vs, store.engine
    expect, raw = [], []
    for f in range(40):
        emb, sp, metas, r = store.file_of(f"f{f}.md", 5 + f % 7)
        ids = vs.store_chunks([(f"{f}:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
        assert len(ids) == len(metas) and not emb.materialized and not sp.materialized
        expect += [(f"f{f}.md", i) for i in range(len(metas))]
        raw.append(r)
    # the host table is complete at once; the engine catches up by itself (linger 20 ms) or when somebody reads
    assert vs.count_by_file("f3.md") == 8 and sum(vs.get_file_chunk_counts().values()) == len(expect)
    assert vs.get_collection_info()["points_count"] == len(expect)
    assert engine.encodes == 0 and sum(engine.batches) == len(expect) and len(engine.batches) < 40
    x = np.concatenate([toy_embed(r[0], r[1]) for r in raw])
    assert np.array_equal(engine.x, ocore.cosine_preprocess(x))
    sp_rows = [row for r in raw for row in toy_tf(r[2], r[3])]
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(engine.sp, sp_rows))
    # search = read-your-writes: a chunk stored a moment ago is found by its own vector
    emb, sp, metas, r = store.file_of("late.md", 3)
    vs.store_chunks([(f"late:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    hit = vs.search(toy_embed(r[0], r[1])[1].tolist(), limit=1)[0]
    assert (hit.metadata.file_path, hit.metadata.chunk_index, hit.text) == ("late.md", 1, "late:1")


def test_looking_at_an_embedding_computes_it_and_the_plain_path_stores_it(store):
    vs, engine = store.vs, store.engine
    emb, sp, metas, r = store.file_of("a.md", 4)
    want = toy_embed(r[0], r[1])
    assert len(emb) == 4 and len(emb[0]) == DIM and engine.encodes == 0
    assert emb[2][5] == float(want[2, 5]) and engine.encodes == 1 and emb.materialized
    assert list(emb[1]) == want[1].tolist() and np.array_equal(np.asarray(emb), want) and emb.tolist() == want.tolist()
    assert np.array_equal(np.array([e for e in emb], np.float32), want) and engine.encodes == 1
    pair = sp[1]
    assert tuple(pair) == (toy_tf(r[2], r[3])[1][0].tolist(), toy_tf(r[2], r[3])[1][1].tolist()) and len(pair) == 2
    vs.store_chunks([(f"a:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    assert engine.batches == [] and np.array_equal(engine.x, ocore.cosine_preprocess(want))  # through upsert, not the fused call


def test_subset_reordered_and_dense_only_stores(store):
    vs, engine = store.vs, store.engine
    emb, sp, metas, r = store.file_of("s.md", 6)
    want = ocore.cosine_preprocess(toy_embed(r[0], r[1]))
    pick = [4, 0, 3]
    vs.store_chunks([(f"s:{i}", emb[i], metas[i]) for i in pick])  # no sparse vectors: dense-only rows
    e0, s0, m0, r0 = store.file_of("s0.md", 2)  # queued right behind it, WITH sparse vectors: a fused call of its own
    vs.store_chunks([(f"s0:{i}", e, m) for i, (e, m) in enumerate(zip(e0, m0))], sparse_vectors=s0)
    vs.flush()
    assert engine.batches == [3, 2] and engine.sp[3] is not None and engine.sp[2] is None
    vs.delete_by_file("s0.md")
    engine.x, engine.sp, engine.live = engine.x[:3], engine.sp[:3], engine.live[:3]  # (back to three rows for the checks below)
    for a in ("folder", "ifolder", "created", "modified"):
        setattr(engine, a, getattr(engine, a)[:3])
    col = vs._col
    del col.ids[3:], col.payload[3:]
    assert np.array_equal(engine.x, want[pick]) and all(row is None for row in engine.sp)  # no sparse vector at all
    # more chunks than sparse vectors -> ordinary path (the reference stores the surplus dense-only)
    emb2, sp2, metas2, r2 = store.file_of("t.md", 3)
    vs.store_chunks([(f"t:{i}", e, m) for i, (e, m) in enumerate(zip(emb2, metas2))], sparse_vectors=list(sp2)[:2])
    assert engine.x.shape[0] == 6 and len(engine.sp[5][0]) == 0 and len(engine.sp[4][0]) > 0 and engine.sp[0] is None
    # references of two different embed_texts calls in one store -> ordinary path, same rows
    e3, s3, m3, r3 = store.file_of("u.md", 2)
    e4, s4, m4, r4 = store.file_of("v.md", 2)
    vs.store_chunks([("u0", e3[0], m3[0]), ("v1", e4[1], m4[1])])
    assert np.array_equal(engine.x[6:], ocore.cosine_preprocess(np.stack([toy_embed(r3[0], r3[1])[0], toy_embed(r4[0], r4[1])[1]])))


def test_delete_and_plain_store_wait_for_queued_rows(store):
    vs, engine = store.vs, store.engine
    engine.delay = 0.05
    for f in range(6):
        # IndexingService deletes a file's chunks before indexing it, every file (indexing.py:281-288): a delete that
        # finds nothing must not wait for the queue, or the write-behind could never batch across files
        assert vs.count_by_file(f"f{f}.md") == 0 and vs.delete_by_file(f"f{f}.md") == 0
        emb, sp, metas, _ = store.file_of(f"f{f}.md", 10)
        vs.store_chunks([(f"{f}:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    assert sum(engine.batches) < 60  # (rows are still queued: the six deletes above did not drain them)
    assert vs.delete_by_file("f5.md") == 10  # its rows were still queued: drained first, then deleted in the engine
    assert engine.count() == (60, 50)
    emb, sp, metas, _ = store.file_of("g.md", 2)
    vs.store_chunks([(f"g:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    plain = np.ones((1, DIM), np.float32)
    vs.store_chunks([("p", plain[0].tolist(), metas[0])])  # a list of floats: direct upsert, but AFTER the queued rows
    assert engine.count() == (63, 53) and np.array_equal(engine.x[62], ocore.cosine_preprocess(plain)[0])


def test_a_failed_fused_call_takes_its_rows_back_and_is_reported(store):
    vs, engine = store.vs, store.engine
    emb, sp, metas, _ = store.file_of("ok.md", 3)
    vs.store_chunks([(f"ok:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    vs.flush()
    engine.fail_next = True
    emb, sp, metas, _ = store.file_of("bad.md", 4)
    vs.store_chunks([(f"bad:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    with pytest.raises(RuntimeError, match="could not be completed"):
        vs.flush()
    assert vs.count_by_file("bad.md") == 0 and vs.count_by_file("ok.md") == 3 and engine.count() == (3, 3)
    # the files whose rows were taken back are named, so that a caller with its own bookkeeping can map the failure
    assert vs.failed_file_paths() == ["bad.md"] and vs.failed_file_paths() == []
    emb, sp, metas, _ = store.file_of("next.md", 2)  # the store keeps working, rows stay aligned
    vs.store_chunks([(f"n:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    vs.flush()
    assert engine.count() == (5, 5) and vs.count_by_file("next.md") == 2


def test_switch_off_and_concurrent_callers(store, monkeypatch):
    vs, engine = store.vs, store.engine
    errors = []

    def worker(t):
        try:
            for f in range(15):
                emb, sp, metas, _ = files[t][f]
                vs.store_chunks([(f"{t}:{f}:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
                if f % 5 == 4:
                    vs.search(np.ones(DIM).tolist(), limit=3)
        except BaseException as e:  # noqa: BLE001
            errors.append(repr(e))

    files = [[store.file_of(f"t{t}f{f}.md", 4) for f in range(15)] for t in range(4)]
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    vs.flush()
    assert not errors and engine.count() == (240, 240)
    # every host row describes the engine row with the same number
    col = vs._col
    for t in range(4):
        for f in range(15):
            rows = col.rows_by_file[f"t{t}f{f}.md"]
            ids, off = files[t][f][3][0], files[t][f][3][1]
            assert np.array_equal(engine.x[rows], ocore.cosine_preprocess(toy_embed(ids, off)))
    monkeypatch.setenv("VOITTA_DEFERRED_INDEXING", "0")
    assert not store.deferred.enabled()


def test_default_mode_keeps_the_reference_failure_contract(store, monkeypatch):
    """VOITTA_DEFERRED_INDEXING unset: the embeddings are still references (no Python floats), but store_chunks makes
    its ONE fused call itself and raises if it fails — where the reference's upsert raises (vector_store.py:311-313),
    so that IndexingService marks the file failed instead of committing it (indexing.py:558-590)."""
    monkeypatch.delenv("VOITTA_DEFERRED_INDEXING", raising=False)
    assert store.deferred.enabled() and not store.deferred.write_behind()
    vs, engine = store.vs, store.engine
    emb, sp, metas, r = store.file_of("a.md", 4)
    vs.store_chunks([(f"a:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    assert engine.batches == [4] and engine.encodes == 0 and engine.count() == (4, 4)   # in the engine when the call returns
    assert not emb.materialized and vs._col.flusher is None                              # no floats, no queue
    assert np.array_equal(engine.x, ocore.cosine_preprocess(toy_embed(r[0], r[1])))
    engine.fail_next = True
    emb, sp, metas, _ = store.file_of("bad.md", 3)
    with pytest.raises(RuntimeError, match="engine said no"):
        vs.store_chunks([(f"bad:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    assert vs.count_by_file("bad.md") == 0 and engine.count() == (4, 4)
    emb, sp, metas, _ = store.file_of("b.md", 2)
    vs.store_chunks([(f"b:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    assert engine.count() == (6, 6) and vs.count_by_file("b.md") == 2


def test_a_search_waits_for_the_stores_before_it_only(store):
    """_drain works on sequence numbers: a search waits until the stores that returned BEFORE it are in the engine,
    not until an indexing thread that keeps queueing has gone quiet; and a failed background store is not raised out
    of an unrelated search."""
    vs, engine = store.vs, store.engine
    engine.delay = 0.02
    emb, sp, metas, r = store.file_of("first.md", 3)
    vs.store_chunks([(f"first:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    stop = threading.Event()
    stored = []

    def feeder():
        f = 0
        while not stop.is_set() and f < 400:
            e2, s2, m2, _ = store.file_of(f"bg{f}.md", 2)
            vs.store_chunks([(f"bg{f}:{i}", e, m) for i, (e, m) in enumerate(zip(e2, m2))], sparse_vectors=s2)
            stored.append(f)
            f += 1
            time.sleep(0.001)

    t = threading.Thread(target=feeder)
    t.start()
    try:
        time.sleep(0.01)
        t0 = time.monotonic()
        hit = vs.search(toy_embed(r[0], r[1])[0].tolist(), limit=1)[0]
        waited = time.monotonic() - t0
        assert hit.metadata.file_path == "first.md"
        assert waited < 1.0 and not stop.is_set() and t.is_alive()   # the feeder is still queueing: the search did not wait for it
    finally:
        stop.set()
        t.join(30)
    vs.flush()
    engine.delay = 0.0
    engine.fail_next = True
    emb, sp, metas, _ = store.file_of("bad.md", 2)
    vs.store_chunks([(f"bad:{i}", e, m) for i, (e, m) in enumerate(zip(emb, metas))], sparse_vectors=sp)
    assert vs.search(toy_embed(r[0], r[1])[0].tolist(), limit=1)[0].metadata.file_path == "first.md"   # no error here
    with pytest.raises(RuntimeError, match="could not be completed"):
        vs.flush()
    assert vs.failed_file_paths() == ["bad.md"]
