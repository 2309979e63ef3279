"""Search while the index is mutated. The reference is hit by MCP worker threads (search), the
indexing thread (store, delete-before-reindex: indexing.py:281-288), the watcher (watcher.py:149-171) and the event
loop (api/routes/folders.py:137-143) at once. Here: four searcher threads and one mutator on ONE VectorStoreService;
every single search result must equal the CPU oracle's answer for a state of the collection that existed while that
search ran — the state before or after each mutation it overlapped, never a mixture — for dense and hybrid queries,
across appends, deletes and compactions (which renumber every row while searches are in flight)."""
import threading
import time

import numpy as np
import pytest

from oracle import core as ocore
from oracle import fusion as ofus
from test_services_gpu import native  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

DIM = 128


def _make(rng, n_files, per_file, first_file=0):
    from voitta_rag_amd.vector_store import ChunkMetadata

    chunks, sparse = [], []
    for f in range(first_file, first_file + n_files):
        for i in range(per_file):
            meta = ChunkMetadata(file_path=f"d{f % 7}/f{f}.md", folder_path=f"d{f % 7}", index_folder=f"d{f % 7}",
                                 file_name=f"f{f}.md", chunk_index=i, total_chunks=per_file, start_char=0, end_char=1,
                                 indexed_at="t")
            chunks.append((f"{f}:{i}", rng.standard_normal(DIM).astype(np.float32), meta))
            m = int(rng.integers(2, 7))
            sparse.append(((rng.choice(200, size=m, replace=False) * 13 + 5).astype(np.int32),
                           rng.uniform(0.3, 2.0, size=m).astype(np.float32)))
    return chunks, sparse


class OracleCollection:
    """The collection as the oracle sees it: rows in append order, a live mask, answers by (file, chunk) key."""

    def __init__(self):
        self.x = np.zeros((0, DIM), np.float32)
        self.sp, self.keys, self.folder = [], [], []
        self.live = np.zeros(0, bool)

    def append(self, chunks, sparse):
        self.x = np.concatenate([self.x, ocore.cosine_preprocess(np.array([c[1] for c in chunks], np.float32))])
        self.sp += [(np.sort(i), v[np.argsort(i, kind="stable")]) for i, v in sparse]
        self.keys += [(c[2].file_path, c[2].chunk_index) for c in chunks]
        self.folder += [c[2].folder_path for c in chunks]
        self.live = np.concatenate([self.live, np.ones(len(chunks), bool)])

    def delete_file(self, fp):
        self.live &= ~np.array([k[0] == fp for k in self.keys])

    def answer(self, q, sq, limit, sw, folder=None):
        mask = self.live & (np.array(self.folder) == folder if folder else True)
        m8 = mask.astype(np.uint8)
        dsc = ocore.dense_scores(ocore.cosine_preprocess(np.asarray(q, np.float32)[None]), self.x)[0]
        if sq is None:
            r, s = ocore.topk(dsc, limit, m8)
            return [(self.keys[i], float(str(np.float32(v)))) for i, v in zip(r, s)]
        rows = [row if self.live[i] else None for i, row in enumerate(self.sp)]
        ssc = ocore.sparse_scores(rows, sq[0], sq[1], self.live)
        dr, ds = ocore.topk(dsc, 3 * limit, m8)
        sr, ss = ocore.topk(ssc, 3 * limit, m8)
        fused = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit, sw, "json")
        return [(self.keys[r], s) for r, s, _ in fused]


def test_searches_see_the_state_before_or_after_every_mutation(native):  # noqa: F811
    from voitta_rag_amd.vector_store import VectorStoreService, get_vector_store

    native(dims=(1, DIM, 4, 256))  # (only the store is used; the collection is 128-dimensional)
    rng = np.random.default_rng(2026)
    vs = get_vector_store()
    ora = OracleCollection()
    base_chunks, base_sparse = _make(rng, 60, 300)  # 18,000 chunks to start with (two-stage search territory)
    for a in range(0, len(base_chunks), 3000):
        vs.store_chunks([(t, v.tolist(), m) for t, v, m in base_chunks[a:a + 3000]], sparse_vectors=base_sparse[a:a + 3000])
    ora.append(base_chunks, base_sparse)

    queries = []
    for j in range(8):
        q = rng.standard_normal(DIM).astype(np.float32)
        sq = ((rng.choice(200, size=3, replace=False) * 13 + 5).astype(np.int32).tolist(), [1.0, 1.0, 1.0])
        queries.append((q, sq if j % 2 else None, 10 if j % 3 else 7, 0.3, "d3" if j == 5 else None))

    # the mutation script and the oracle's answer for every state
    script = []
    next_file = 60
    for step in range(18):
        kind = ("store", "delete", "store", "delete", "store", "compact")[step % 6]
        if kind == "store":
            c, s = _make(rng, 2, 120, next_file)
            next_file += 2
            script.append(("store", c, s))
        elif kind == "delete":
            f = (11 * step + 3) % next_file
            script.append(("delete", f"d{f % 7}/f{f}.md"))
        else:
            script.append(("compact",))
    expected = [[ora.answer(*q) for q in queries]]
    for op in script:
        if op[0] == "store":
            ora.append(op[1], op[2])
        elif op[0] == "delete":
            ora.delete_file(op[1])
        expected.append([ora.answer(*q) for q in queries])
    assert any(expected[i] != expected[i + 1] for i in range(len(script))), "the script must change some answers"

    spans = []  # (t_begin, t_end) of every mutation, in order
    stop = threading.Event()
    failures, records = [], []

    def mutator():
        try:
            time.sleep(0.05)
            for op in script:
                t0 = time.perf_counter()
                if op[0] == "store":
                    vs.store_chunks([(t, v.tolist(), m) for t, v, m in op[1]], sparse_vectors=op[2])
                elif op[0] == "delete":
                    vs.delete_by_file(op[1])
                else:
                    vs.compact()
                spans.append((t0, time.perf_counter()))
                time.sleep(0.01)
        except BaseException as e:  # noqa: BLE001
            failures.append(("mutator", repr(e)))
        finally:
            stop.set()

    def searcher(seed):
        local = VectorStoreService()  # a fresh service object per thread, as api/routes/folders.py:137-143 makes one
        r = np.random.default_rng(seed)
        try:
            while not stop.is_set():
                j = int(r.integers(len(queries)))
                q, sq, limit, sw, folder = queries[j]
                t0 = time.perf_counter()
                got = local.search(q.tolist(), limit=limit, sparse_query=sq, sparse_weight=sw, folder_filter=folder)
                t1 = time.perf_counter()
                records.append((j, t0, t1, [((c.metadata.file_path, c.metadata.chunk_index), c.score) for c in got]))
        except BaseException as e:  # noqa: BLE001
            failures.append(("searcher", repr(e)))
            stop.set()

    threads = [threading.Thread(target=mutator)] + [threading.Thread(target=searcher, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not failures, failures
    assert len(spans) == len(script) and len(records) > 4 * len(script), (len(spans), len(records))
    overlapped = 0
    for j, t0, t1, got in records:
        lo = sum(1 for b, e in spans if e <= t0)  # mutations certainly applied before the search began
        hi = sum(1 for b, e in spans if b < t1)   # mutations that may have been applied before it ended
        overlapped += hi > lo
        assert any(got == expected[s][j] for s in range(lo, hi + 1)), (j, lo, hi, got[:3], expected[lo][j][:3])
    assert overlapped > 0  # some searches really ran beside a mutation
    # and afterwards: the final state, from every thread's point of view
    for j, q in enumerate(queries):
        got = vs.search(q[0].tolist(), limit=q[2], sparse_query=q[1], sparse_weight=q[3], folder_filter=q[4])
        assert [((c.metadata.file_path, c.metadata.chunk_index), c.score) for c in got] == expected[-1][j]


def test_concurrent_engine_searches_equal_serial_ones(gpu):
    """Eight threads on one engine (four lanes): every answer equals the single-threaded one, bit for bit."""
    from voitta_rag_amd import Engine

    rng = np.random.default_rng(3)
    n = 30000
    x = rng.standard_normal((n, DIM)).astype(np.float32)
    sp = [((rng.choice(300, size=4, replace=False) * 7 + 2).astype(np.int32), rng.uniform(0.5, 2.0, size=4).astype(np.float32))
          for _ in range(n)]
    e = Engine(DIM, initial_rows=n)
    e.upsert(x, sparse=sp)
    qs = rng.standard_normal((64, DIM)).astype(np.float32)
    sq = [((rng.choice(300, size=3, replace=False) * 7 + 2).astype(np.int32), np.ones(3, np.float32)) for _ in range(64)]
    serial = [(e.search_dense(qs[i:i + 1], 10)[0], e.search_hybrid(qs[i], sq[i][0], sq[i][1], 10, 0.2), e.search_sparse(sq[i][0], sq[i][1], 10))
              for i in range(64)]
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                for i in range(t, 64, 8):
                    d = e.search_dense(qs[i:i + 1], 10)[0]
                    h = e.search_hybrid(qs[i], sq[i][0], sq[i][1], 10, 0.2)
                    s = e.search_sparse(sq[i][0], sq[i][1], 10)
                    ok = (np.array_equal(d[0], serial[i][0][0]) and np.array_equal(d[1], serial[i][0][1])
                          and all(np.array_equal(a, b) for a, b in zip(h, serial[i][1]))
                          and np.array_equal(s[0], serial[i][2][0]) and np.array_equal(s[1], serial[i][2][1]))
                    if not ok:
                        errors.append((t, rep, i))
        except BaseException as ex:  # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:5]
    # a batched search beside single ones
    batch = e.search_dense(qs, 10)
    for i in range(64):
        assert np.array_equal(batch[i][0], serial[i][0][0]) and np.array_equal(batch[i][1], serial[i][0][1])
    e.close()
