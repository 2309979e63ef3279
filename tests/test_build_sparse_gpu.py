"""a8 — the offline BM25 rebuild (reference: scripts/build_sparse_vectors.py:73-245): a collection stored
dense-only is migrated into a new one with sparse vectors; ids, payloads and dense vectors carry over,
points without text get no sparse vector, and the migrated collection answers hybrid queries exactly
like a collection that was indexed with sparse vectors from the start."""
import os

import numpy as np
import pytest

from test_services_gpu import native, _texts  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _answers(vs, emb, sp, queries):
    out = []
    for q in queries:
        qv = emb.embed_query(q)
        for kw in ({}, {"sparse_query": sp.embed_query(q), "sparse_weight": 0.4}, {"sparse_query": sp.embed_query(q), "sparse_weight": 1.0},
                   {"sparse_query": sp.embed_query(q), "sparse_weight": 0.4, "include_folders": ["b"]}):
            out.append([(r.id if kw is None else (r.metadata.file_path, r.metadata.chunk_index), r.score)
                        for r in vs.search(qv, limit=8, **kw)])
    return out


def _index(vs, emb, sp, texts, with_sparse):
    from voitta_rag_amd.vector_store import ChunkMetadata

    embeddings = emb.embed_texts(texts)
    chunks = [(t, e, ChunkMetadata(file_path=f"{'ab'[i % 2]}/f{i // 10}.md", folder_path="ab"[i % 2], index_folder="ab"[i % 2],
                                   file_name=f"f{i // 10}.md", chunk_index=i % 10, total_chunks=10, start_char=0, end_char=len(t),
                                   indexed_at="t", source_modified_at=1_700_000_000 + i))
              for i, (t, e) in enumerate(zip(texts, embeddings))]
    return vs.store_chunks(chunks, sparse_vectors=sp.embed_texts(texts) if with_sparse else None)


def test_rebuild_matches_a_collection_indexed_with_sparse_vectors(native, tmp_path, monkeypatch):  # noqa: F811
    from voitta_rag_amd import store_registry
    from voitta_rag_amd.build_sparse import build_sparse_vectors
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import get_vector_store

    rng = np.random.default_rng(21)
    texts = _texts(rng, 230)
    queries = ["vector database index", "running happily", "memory bandwidth kernel", "hybrid fusion ranking"]

    # the answer key: sparse vectors from the start; a few rows deleted so that the source has tombstones
    native("mini-a")
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    _index(vs, emb, sp, texts, with_sparse=True)
    vs.delete_by_file("a/f3.md")
    want = _answers(vs, emb, sp, queries)
    want_count = vs.client.count()[1]

    # the source: dense only (what the reference's collection looked like before the BM25 migration)
    index_dir = tmp_path / "index"
    monkeypatch.setenv("VOITTA_INDEX_DIR", str(index_dir))
    native("mini-b")
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    ids = _index(vs, emb, sp, texts, with_sparse=False)
    vs.delete_by_file("a/f3.md")
    assert vs.client.sparse_stats(np.array(sp.embed_query("vector")[0], np.int32))[1] == 0  # no sparse points yet
    dense_before = _answers(vs, emb, sp, queries)[0::4]

    dry = build_sparse_vectors(dry_run=True)
    assert dry["processed"] == want_count and dry["inserted"] == 0 and vs.client.count()[1] == want_count

    stats = build_sparse_vectors(batch_size=64, switch=True)
    assert stats["target"] == "voitta_documents_v2" and stats["processed"] == stats["inserted"] == want_count
    assert stats["skipped"] == 0 and stats["rate"] > 0
    assert sorted(os.listdir(index_dir)) == ["voitta_documents_v2.g1.payload.jsonl", "voitta_documents_v2.g1.vrindex",
                                             "voitta_documents_v2.meta.json"]

    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    assert vs.client.count() == (want_count, want_count)  # tombstones were not copied
    got = _answers(vs, emb, sp, queries)
    assert got[0::4] == dense_before           # dense answers untouched by the migration
    assert got == want                         # and everything equals the collection built with BM25 from the start
    kept = [i for i in ids if i in vs._col.row_of]
    assert len(kept) == want_count             # point ids carried over
    # scripts/sync_qdrant_stats.py:29-81: the per-file aggregate the reference scrolls the collection for
    per_file = vs.scan_file_stats()
    assert sum(f["chunk_count"] for f in per_file.values()) == want_count and "a/f3.md" not in per_file
    assert per_file["b/f0.md"] == {"folder_path": "b", "index_folder": "b", "chunk_count": 5, "indexed_at": "t"}
    assert {fp: f["chunk_count"] for fp, f in per_file.items()} == vs.get_file_chunk_counts()

    # the reference's switch-over: a restart with QDRANT_COLLECTION=<target> serves the saved target
    monkeypatch.setenv("QDRANT_COLLECTION", "voitta_documents_v2")
    native("mini-c")
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    assert vs.collection_name == "voitta_documents_v2" and vs.client.count() == (want_count, want_count)
    assert _answers(vs, emb, sp, queries) == want
    store_registry.reset()


def test_points_without_text_get_no_sparse_vector(native):  # noqa: F811
    from voitta_rag_amd.build_sparse import build_sparse_vectors
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import get_vector_store

    native()
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    texts = ["vector database", "sparse dense hybrid", "retrieval query", "index folder", "kernel memory"]
    _index(vs, emb, sp, texts, with_sparse=False)
    for r in (1, 2):  # payloads whose text was lost (build_sparse_vectors.py:158-165)
        vs._col.payload[r]["text"] = ""
    stats = build_sparse_vectors(switch=True)
    assert (stats["processed"], stats["inserted"], stats["skipped"]) == (5, 5, 2)
    vs = get_vector_store()
    _, n_points = vs.client.sparse_stats(np.zeros(1, np.int32))
    assert n_points == 3  # the IDF's N counts the three points that carry a sparse vector


def test_migrated_collection_against_the_bm25_oracle(native):  # noqa: F811
    """The target of the migration checked against oracle/bm25.py + oracle_core, NOT against a second HIP
    collection (scripts/build_sparse_vectors.py:153-194): every point with text carries exactly the sparse vector
    the oracle computes from the STORED text, a point without text carries none and does not count towards the
    IDF's N, document frequencies follow, sparse and hybrid rankings and their f32 / f64 scores are the oracle's,
    and ids, payloads and dense rows carry over bit for bit."""
    from oracle import bm25 as obm
    from oracle import core as ocore
    from oracle import fusion as ofus
    from voitta_rag_amd.build_sparse import build_sparse_vectors
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import get_vector_store

    native()
    rng = np.random.default_rng(33)
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    texts = _texts(rng, 140)
    ids = _index(vs, emb, sp, texts, with_sparse=False)
    lost = [4, 17, 18, 99]  # payloads whose text was lost: skipped by the reference (:158-165)
    for r in lost:
        vs._col.payload[r]["text"] = ""
    vs.delete_by_file("b/f1.md")  # rows 11, 13, 15, 17, 19: tombstones in the source are not copied
    src_rows = list(vs._col.live_rows())
    assert len(src_rows) == len(texts) - 5
    dense_before = vs.client.get_dense(np.array(src_rows, np.int64))
    payload_before = [dict(vs._col.payload[r]) for r in src_rows]
    ids_before = [vs._col.ids[r] for r in src_rows]

    stats = build_sparse_vectors(batch_size=37, switch=True)  # odd batch size: runs with/without text cross batches
    n = len(src_rows)
    n_lost = sum(1 for r in src_rows if r in lost)
    assert (stats["processed"], stats["inserted"], stats["skipped"]) == (n, n, n_lost)

    emb, vs = get_embedding_service(), get_vector_store()  # (the old services were bound to the closed source engine)
    e = vs.client
    assert e.count() == (n, n)
    # carry-over (:176-194): ids, payloads, dense bits, in the source's order
    assert [vs._col.ids[i] for i in range(n)] == ids_before and set(ids_before) <= set(ids)
    assert [vs._col.payload[i] for i in range(n)] == payload_before
    assert np.array_equal(e.get_dense(np.arange(n)).view(np.uint32), dense_before.view(np.uint32))

    # the oracle's sparse rows from the STORED texts (f64 tf -> f32 as stored); no text -> no sparse vector
    want_rows = []
    for p in payload_before:
        if not p["text"]:
            want_rows.append(None)
            continue
        m = obm.term_frequency(obm.stems(p["text"]))
        idx = np.array(sorted(m), np.int32)
        want_rows.append((idx, np.array([m[int(t)] for t in idx], np.float64).astype(np.float32)))
    df_want, n_points = ocore.document_frequencies(want_rows)
    assert n_points == n - n_lost
    probe = np.array(sorted(df_want) + [123456789], np.int32)
    df_got, n_got = e.sparse_stats(probe)
    assert n_got == n_points
    assert [int(v) for v in df_got] == [df_want.get(int(t), 0) for t in probe]

    stored = e.get_dense(np.arange(n))
    for qtext in ("vector database index", "running happily", "memory bandwidth kernel", "hybrid fusion ranking", "chunk"):
        qi, qv = obm.query_embed(qtext)
        want = ocore.sparse_scores(want_rows, qi, qv)
        for k in (10, 30):
            wr, ws = ocore.topk(want, k)
            gr, gs = e.search_sparse(qi, qv, k)
            assert np.array_equal(gr, wr), qtext
            assert np.array_equal(gs.view(np.uint32), ws.view(np.uint32)), qtext
        assert not any(want_rows[int(r)] is None for r in wr)  # a point without text never answers a sparse query
        q = np.asarray(emb.embed_query(qtext), np.float32)
        dsc = ocore.dense_scores(ocore.cosine_preprocess(q[None]), stored)[0]
        dr, ds = ocore.topk(dsc, 30)
        sr, ss = ocore.topk(want, 30)
        fused = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), 10, 0.4, "json")
        got = vs.search(q.tolist(), limit=10, sparse_query=(qi, qv), sparse_weight=0.4)
        assert [vs._col.row_of[c.id] for c in got] == [r for r, _, _ in fused]
        assert [c.score for c in got] == [s for _, s, _ in fused]
