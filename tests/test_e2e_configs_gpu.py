"""BASELINE configs[0] and configs[1] at their own shape, end to end through the drop-in services.

configs[0] (SURVEY.md §8d cfg 1, "the parity/oracle run"): 1k synthetic text chunks, all-MiniLM-L6-v2 SHAPE
(L6 H384 12 heads I1536, mean pooling; seeded weights and a synthetic vocab — there is no checkpoint offline),
encode -> D=384 store -> 100 hybrid top-10 queries with sparse_weight 0.1, the call sequence of
indexing.py:527-560 and mcp_server.py:469-485. Checked against the oracle at every boundary:
  * embeddings: |1 - cos| < 1e-5 against oracle/bert.py in f64 (north_star: 1e-4),
  * stored rows: bit-equal to the oracle's cosine preprocessing of the vectors handed to store_chunks,
  * every query: ranked point ids and fused f64 scores equal oracle dense top-30 + oracle BM25/IDF sparse
    top-30 + the restated _hybrid_search, bit for bit,
  * recall@10 of the whole pipeline against the SAME pipeline run entirely by the oracle in f64
    (its own embeddings for chunks and queries): >= 0.99 (north_star).
configs[1]: 100k x 384 dense-only rows — size-independent properties (two code paths agree, merge of halves
equals the whole, winners carry the oracle's exact scores)."""
import numpy as np
import pytest

from oracle import bert as obert
from oracle import bm25 as obm
from oracle import core as ocore
from oracle import fusion as ofus
from test_services_gpu import WORDS, _oracle_embed, native  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

MINILM = (6, 384, 12, 1536)  # all-MiniLM-L6-v2: layers, hidden, heads, intermediate (SURVEY.md §8 model table)


def _chunks(rng, n):
    out = []
    for _ in range(n):
        words = rng.choice(WORDS, size=int(rng.integers(3, 28)))
        out.append(" ".join(words) + str(rng.choice([".", "!", "?", ""])))
    return out


def test_config0_1k_chunks_minilm_hybrid_top10_against_the_oracle(native):  # noqa: F811
    path, shape, w, vocab = native("minilm-shape", "mean", dims=MINILM, max_seq=32)
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, get_vector_store

    rng = np.random.default_rng(1234)
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    assert emb.dimension == 384
    texts = _chunks(rng, 1000)
    queries = [" ".join(rng.choice(WORDS, size=int(rng.integers(4, 9)))) for _ in range(100)]

    # ---- index, file by file like IndexingService (indexing.py:527-560): 40 files of 25 chunks
    ids, handed = [], []
    for f in range(40):
        part = texts[25 * f: 25 * (f + 1)]
        embeddings = emb.embed_texts(part)
        sparse_vectors = sp.embed_texts(part)
        metas = [ChunkMetadata(file_path=f"docs/f{f}.md", folder_path="docs", index_folder="docs", file_name=f"f{f}.md",
                               chunk_index=i, total_chunks=len(part), start_char=0, end_char=len(t), indexed_at="t")
                 for i, t in enumerate(part)]
        ids += vs.store_chunks(list(zip(part, embeddings, metas)), sparse_vectors=sparse_vectors)
        handed += embeddings
    n = len(texts)
    assert vs.get_collection_info()["points_count"] == n

    # ---- embeddings against the f64 oracle
    want_emb = _oracle_embed(texts + ["x " * 40], shape, w, vocab, "mean")[:n]  # (+ one long text: truncation exercised)
    got_emb = np.asarray(handed, np.float32)
    cos = (got_emb * want_emb).sum(1) / np.linalg.norm(got_emb, axis=1) / np.linalg.norm(want_emb, axis=1)
    assert np.max(np.abs(1 - cos)) < 1e-5, np.max(np.abs(1 - cos))

    # ---- stored rows: the oracle's Qdrant-side preprocessing of exactly what store_chunks was given
    xh = ocore.cosine_preprocess(got_emb)
    assert np.array_equal(vs.client.get_dense(np.arange(n)).view(np.uint32), xh.view(np.uint32))
    sp_rows = []
    for t in texts:
        m = obm.term_frequency(obm.stems(t))
        idx = np.array(sorted(m), np.int32)
        sp_rows.append((idx, np.array([m[int(i)] for i in idx], np.float64).astype(np.float32)))

    # ---- 100 hybrid top-10 queries (mcp_server.py:469-485), ids and fused scores bit for bit
    want_q = _oracle_embed(queries + ["x " * 40], shape, w, vocab, "mean")[:100]
    xo = (want_emb / np.linalg.norm(want_emb, axis=1, keepdims=True))
    hits = 0
    worst_q = 0.0
    for qi_, q in enumerate(queries):
        qv = emb.embed_query(q)
        sq = sp.embed_query(q)
        assert sq == obm.query_embed(q)
        worst_q = max(worst_q, abs(1 - float(np.dot(qv, want_q[qi_]) / np.linalg.norm(qv) / np.linalg.norm(want_q[qi_]))))
        got = vs.search(qv, limit=10, sparse_query=sq, sparse_weight=0.1)
        dsc = ocore.dense_scores(ocore.cosine_preprocess(np.asarray([qv], np.float32)), xh)[0]
        d64 = xo @ (want_q[qi_] / np.linalg.norm(want_q[qi_]))  # the all-oracle pipeline: f64 embeddings on both sides
        if not sq[0]:  # every word a stop word: the reference takes its dense-only branch (vector_store.py:599-619)
            dr, ds = ocore.topk(dsc, 10)
            assert [c.id for c in got] == [ids[r] for r in dr], q
            assert [c.score for c in got] == [float(str(np.float32(s))) for s in ds], q
            hits += len(set(np.lexsort((np.arange(n), -d64))[:10].tolist()) & set(dr.tolist()))
            continue
        dr, ds = ocore.topk(dsc, 30)
        ssc = ocore.sparse_scores(sp_rows, sq[0], sq[1])
        sr, ss = ocore.topk(ssc, 30)
        want = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), 10, 0.1, "json")
        assert [c.id for c in got] == [ids[r] for r, _, _ in want], q
        assert [c.score for c in got] == [s for _, s, _ in want], q
        o_dr = np.lexsort((np.arange(n), -d64))[:30]
        ref = ofus.hybrid_fuse([(int(r), float(d64[r])) for r in o_dr], list(zip(sr.tolist(), ss.tolist())), 10, 0.1, "exact")
        hits += len({r for r, _, _ in ref} & {r for r, _, _ in want})
    assert worst_q < 1e-5
    assert hits / 1000.0 >= 0.99, hits  # recall@10 against the all-f64 pipeline


@pytest.fixture(scope="module")
def corpus_100k(gpu):
    from voitta_rag_amd import Engine

    rng = np.random.default_rng(100)
    n, dim = 100_000, 384
    x = rng.standard_normal((n, dim), dtype=np.float32)
    e2 = Engine(dim, initial_rows=n)            # two-stage (int8 shadow) search
    e1 = Engine(dim, initial_rows=n, prefilter=False)  # one-stage exact f32 scan
    for a in range(0, n, 25_000):
        e2.upsert(x[a:a + 25_000])
        e1.upsert(x[a:a + 25_000])
    yield x, e1, e2
    e1.close()
    e2.close()


def test_config1_100k_x384_dense_top10_properties(corpus_100k):
    """configs[1]: 100k MiniLM-width rows, dense-only top-10. Full-size, so no oracle scan of every query:
    (i) the two-stage and the one-stage engines agree bit for bit, (ii) results are sorted and free of
    duplicates, (iii) the winners carry exactly the oracle's f32 scores, (iv) the merge of two halves' top-10
    lists equals the whole's, (v) a full oracle scan for a handful of queries."""
    from voitta_rag_amd import Engine

    x, e1, e2 = corpus_100k
    n, dim = x.shape
    rng = np.random.default_rng(7)
    q = rng.standard_normal((40, dim), dtype=np.float32)
    q[3] = x[77]
    xh = ocore.cosine_preprocess(x)
    qh = ocore.cosine_preprocess(q)
    halves = []
    for a, b in ((0, n // 2), (n // 2, n)):
        h = Engine(dim, initial_rows=b - a)
        h.upsert(x[a:b])
        halves.append((a, h))
    for i in range(q.shape[0]):
        r2, s2 = e2.search_dense(q[i:i + 1], 10)[0]
        r1, s1 = e1.search_dense(q[i:i + 1], 10)[0]
        assert np.array_equal(r1, r2) and np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
        assert len(set(r2.tolist())) == 10
        keys = [(-float(s), int(r)) for r, s in zip(r2, s2)]
        assert keys == sorted(keys)
        want_s = ocore.dense_scores(qh[i:i + 1], xh[r2])[0]
        assert np.array_equal(want_s.view(np.uint32), s2.view(np.uint32))
        merged = []
        for a, h in halves:
            rr, ss = h.search_dense(q[i:i + 1], 10)[0]
            merged += [(-float(s), int(r) + a) for r, s in zip(rr, ss)]
        assert sorted(merged)[:10] == keys
        if i < 6:
            wr, ws = ocore.topk(ocore.dense_scores(qh[i:i + 1], xh)[0], 10)
            assert np.array_equal(wr, r2) and np.array_equal(ws.view(np.uint32), s2.view(np.uint32))
    assert e2.stats()["two_stage"] >= 40 and e2.stats()["fallback"] == 0
    for _, h in halves:
        h.close()
