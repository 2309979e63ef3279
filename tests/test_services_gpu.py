"""The three drop-in services end to end on the GPU, written like tests of the reference's own
services would read: EmbeddingService / SparseEmbeddingService / VectorStoreService with the
reference's signatures (src/voitta/services/{embedding,sparse_embedding,vector_store}.py), checked
against the oracles. The checkpoint directory is synthetic (seeded weights + synthetic vocab.txt)."""
import json
import os

import numpy as np
import pytest

from oracle import bert as obert
from oracle import bm25 as obm
from oracle import core as ocore
from oracle import fusion as ofus

pytestmark = pytest.mark.gpu

WORDS = ("vector database index retrieval query embedding sparse dense hybrid fusion ranking chunk document folder "
         "search engine kernel memory bandwidth wavefront matrix tile running jumped happily relational the of and "
         "to in is it that was for on are as with they be at one have this from passage").split()


def _vocab():
    v = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789.,!?:;'-")
    v += ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    v += WORDS + ["##ing", "##ed", "##s", "##ly", "##tion"]
    return list(dict.fromkeys(v))


def _checkpoint(tmp_path, name, pooling, seed=5, dims=(2, 128, 4, 256), max_seq=32):
    d = tmp_path / name
    (d / "1_Pooling").mkdir(parents=True)
    (d / "2_Normalize").mkdir()
    vocab = _vocab()
    shape = obert.BertShape(*dims, vocab=len(vocab), max_pos=max(64, max_seq))
    (d / "config.json").write_text(json.dumps({
        "model_type": "bert", "hidden_size": shape.hidden, "num_hidden_layers": shape.layers,
        "num_attention_heads": shape.heads, "intermediate_size": shape.intermediate, "vocab_size": shape.vocab,
        "max_position_embeddings": shape.max_pos, "type_vocab_size": 2, "layer_norm_eps": 1e-12, "hidden_act": "gelu"}))
    (d / "modules.json").write_text(json.dumps([
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
        {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}]))
    (d / "1_Pooling" / "config.json").write_text(json.dumps({
        "word_embedding_dimension": shape.hidden, "pooling_mode_cls_token": pooling == "cls",
        "pooling_mode_mean_tokens": pooling == "mean"}))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": max_seq, "do_lower_case": True}))
    (d / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    w = obert.random_weights(shape, seed)
    from safetensors.numpy import save_file

    save_file({("bert." + k): v for k, v in w.items()}, str(d / "model.safetensors"))
    return str(d), shape, w, vocab


@pytest.fixture
def native(monkeypatch, tmp_path, gpu):
    from voitta_rag_amd import config, embedding, sparse_embedding, store_registry, vector_store

    def setup(name="mini-model", pooling="mean", **kw):
        path, shape, w, vocab = _checkpoint(tmp_path, name, pooling, **kw)
        monkeypatch.setenv("EMBEDDING_MODEL", path)
        monkeypatch.setenv("EMBEDDING_DIMENSION", str(shape.hidden))
        config.get_settings.cache_clear()
        store_registry.reset()
        embedding._embedding_service = None
        sparse_embedding._sparse_embedding_service = None
        vector_store._vector_store = None
        return path, shape, w, vocab

    yield setup
    store_registry.reset()
    config.get_settings.cache_clear()


def _texts(rng, n):
    return [" ".join(rng.choice(WORDS, size=int(rng.integers(1, 45)))) + rng.choice([".", "!", " running?", ""]) for _ in range(n)]


def _oracle_embed(texts, shape, w, vocab, pooling):
    from voitta_rag_amd.embedding import build_wordpiece_tokenizer

    tok = build_wordpiece_tokenizer(vocab, True)
    tok.enable_truncation(max_length=32)
    seqs = [np.array(e.ids, np.int32) for e in tok.encode_batch(texts)]
    assert max(len(s) for s in seqs) == 32  # truncation exercised
    return obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64)


@pytest.mark.parametrize("pooling", ["mean", "cls"])
def test_embedding_service(native, pooling):
    path, shape, w, vocab = native("mini-" + pooling, pooling)
    from voitta_rag_amd.embedding import get_embedding_service

    svc = get_embedding_service()
    assert svc.dimension == shape.hidden and svc.embed_texts([]) == []
    rng = np.random.default_rng(0)
    texts = _texts(rng, 37) + ["Héllo, naïve wörld — running & jumped!", "x"]
    got = np.array(svc.embed_texts(texts))
    assert isinstance(svc.embed_texts(texts[:2]), list) and isinstance(svc.embed_texts(texts[:2])[0][0], float)
    want = _oracle_embed(texts, shape, w, vocab, pooling)
    cos = (got * want).sum(1) / np.linalg.norm(got, axis=1)
    assert np.max(np.abs(1 - cos)) < 1e-5
    one = np.array(svc.embed_text(texts[3]))
    # one text alone takes the few-token kernels, the batch of 39 the large-batch ones (LayerNorm folded into
    # the GEMMs, different summation orders): the same embedding up to the f16 mode's rounding, not the same bits
    assert one.shape == (shape.hidden,) and abs(1.0 - float(np.dot(one, got[3]))) < 1e-6
    assert np.max(np.abs(one - got[3])) < 1e-4
    assert np.array_equal(np.array(svc.embed_query(texts[3])), one)  # no e5 in the name: no prefixes


def test_e5_prefixes(native):
    path, shape, w, vocab = native("my-e5-mini", "mean")
    from voitta_rag_amd.embedding import get_embedding_service

    svc = get_embedding_service()
    t = "vector database"
    want_p = _oracle_embed(["passage: " + t, " ".join(WORDS)], shape, w, vocab, "mean")[0]
    want_q = _oracle_embed(["query: " + t, " ".join(WORDS)], shape, w, vocab, "mean")[0]
    for got, want in ((svc.embed_text(t), want_p), (svc.embed_texts([t])[0], want_p), (svc.embed_query(t), want_q)):
        assert abs(1 - float(np.dot(got, want))) < 1e-5


def test_full_flow_like_indexing_service_and_mcp_search(native):
    """indexing.py:527-560 then mcp_server.py:469-485, on the native singletons."""
    path, shape, w, vocab = native()
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, VectorStoreService, get_vector_store

    rng = np.random.default_rng(1)
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    files = {"docs/a.md": ("docs", "docs"), "docs/sub/b.md": ("docs/sub", "docs"), "notes/c.txt": ("notes", "notes"),
             "d.txt": ("", "")}
    all_chunks, all_sparse, all_ids, all_meta = [], [], [], []
    for fi, (fp, (folder, index_folder)) in enumerate(files.items()):
        texts = _texts(rng, 20 + fi)
        embeddings = emb.embed_texts(texts)
        sparse_vectors = sp.embed_texts(texts)
        metas = [ChunkMetadata(file_path=fp, folder_path=folder, index_folder=index_folder, file_name=os.path.basename(fp),
                               chunk_index=i, total_chunks=len(texts), start_char=i * 10, end_char=i * 10 + 9,
                               indexed_at="2026-01-01T00:00:00", source_modified_at=(1_700_000_000 + fi * 1000 + i) if fi != 2 else None,
                               source_url="https://x/doc" if fi == 0 else None,
                               source_page_count=7 if fi == 1 else None) for i in range(len(texts))]
        ids = vs.store_chunks(list(zip(texts, embeddings, metas)), sparse_vectors=sparse_vectors)
        assert len(ids) == len(set(ids)) == len(texts)
        all_chunks += texts
        all_sparse += sparse_vectors
        all_ids += ids
        all_meta += metas
    n = len(all_chunks)
    assert vs.get_collection_info()["points_count"] == n
    assert vs.count_by_file("docs/a.md") == 20 and vs.count_by_file("nope") == 0
    assert vs.count_chunks_for_files(["docs/a.md", "nope", "d.txt"]) == {"docs/a.md": 20, "d.txt": 23}
    # "" -> prefix "" -> str.startswith("") is always true in the reference (vector_store.py:789,805): everything
    assert vs.count_chunks_for_folder("docs") == (2, 41) and vs.count_chunks_for_folder("") == (4, 86)
    assert vs.get_folder_stats_batch(["docs", "docs/sub", "zzz"]) == {"docs": (2, 41), "docs/sub": (1, 21), "zzz": (0, 0)}
    assert vs.get_stored_page_count("docs/sub/b.md") == 7 and vs.get_stored_page_count("docs/a.md") is None
    assert vs.get_file_paths_by_index_folder("docs") == {"docs/a.md", "docs/sub/b.md"}
    assert [c.metadata.chunk_index for c in vs.get_chunks_by_range("docs/a.md", 3, 6)] == [3, 4, 5, 6]
    assert len(vs.find_by_source_url("https://x/doc")) == 20
    assert vs.get_file_chunk_counts("docs/") == {"docs/a.md": 20, "docs/sub/b.md": 21}

    # the oracle's view of what is stored
    stored = ocore.cosine_preprocess(np.array([emb.embed_texts([t])[0] for t in all_chunks[:3]], np.float32))
    assert stored.shape[0] == 3  # (smoke: single-text embeds work while the store holds rows)
    dense_all = np.array(emb.embed_texts(all_chunks), np.float32)
    xh = ocore.cosine_preprocess(dense_all)
    sp_rows = [(np.array(i, np.int32), np.array(v, np.float64).astype(np.float32)) for i, v in all_sparse]
    folder_of = np.array([m.folder_path for m in all_meta])
    live = np.ones(n, bool)

    def check(query, limit, sparse_weight, mask, **kw):
        qv = emb.embed_query(query)
        sq = sp.embed_query(query)
        got = VectorStoreService().search(qv, limit=limit, sparse_query=sq, sparse_weight=sparse_weight, **kw)
        m8 = (mask & live).astype(np.uint8)
        dsc = ocore.dense_scores(ocore.cosine_preprocess(np.array([qv], np.float32)), xh)[0]
        if sq[0]:
            dr, ds = ocore.topk(dsc, 3 * limit, m8)
            ssc = ocore.sparse_scores([r if live[i] else None for i, r in enumerate(sp_rows)], sq[0], sq[1], live)
            sr, ss = ocore.topk(ssc, 3 * limit, m8)
            want = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit,
                                    sparse_weight)
            assert [c.id for c in got] == [all_ids[r] for r, _, _ in want]
            assert [c.score for c in got] == [s for _, s, _ in want]
        else:
            dr, ds = ocore.topk(dsc, limit, m8)
            assert [c.id for c in got] == [all_ids[r] for r in dr]
            assert [c.score for c in got] == [float(str(np.float32(s))) for s in ds]
        for c in got:
            r = all_ids.index(c.id)
            assert c.text == all_chunks[r] and c.metadata.file_path == all_meta[r].file_path
        return got

    everything = np.ones(n, bool)
    check("vector database retrieval", 10, 0.1, everything)
    check("the of and", 5, 0.1, everything)  # sparse query empty -> dense-only branch (:599-619)
    check("hybrid fusion ranking", 7, 0.5, folder_of == "docs", folder_filter="docs")
    check("kernel memory", 10, 0.1, np.isin(folder_of, ["docs/sub", "notes"]), include_folders=["docs/sub", "notes", "ghost"])
    check("chunk document", 10, 0.1, ~np.isin(folder_of, ["docs"]), exclude_folders=["docs"], exclude_index_folders=["nothing"])
    mod = np.array([m.source_modified_at if m.source_modified_at is not None else -1 for m in all_meta])
    check("search engine", 10, 0.1, (mod >= 1_700_001_000), date_start=1_700_001_000)
    assert VectorStoreService().search(emb.embed_query("x"), folder_filter="ghost") == []

    # re-index one file: delete + store again (indexing.py:281-288)
    assert vs.delete_by_file("docs/a.md") == 20 and vs.delete_by_file("docs/a.md") == 0
    live[:20] = False
    check("vector database retrieval", 10, 0.1, everything)
    assert vs.delete_by_folder("notes") == 22
    live[[i for i, m in enumerate(all_meta) if m.folder_path == "notes"]] = False
    assert vs.delete_by_index_folder("docs") == 21
    live[[i for i, m in enumerate(all_meta) if m.index_folder == "docs"]] = False
    check("running happily", 10, 0.3, everything)
    vs.set_file_acl("d.txt", ["a@b.c"])
    assert all(c.metadata.allowed_users == ["a@b.c"] for c in vs.get_chunks_by_range("d.txt", 0, 99))
    assert vs.get_collection_info()["points_count"] == 23


def test_collection_survives_a_restart(native, tmp_path, monkeypatch):
    """save() -> new process state (registry reset) -> a fresh VectorStoreService finds everything
    again, like the reference re-attaching to Qdrant's volume (vector_store.py:75-115)."""
    path, shape, w, vocab = native()
    from voitta_rag_amd import config, store_registry, vector_store
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, VectorStoreService

    rng = np.random.default_rng(11)
    emb, sp = get_embedding_service(), get_sparse_embedding_service()
    vs = VectorStoreService()
    texts = _texts(rng, 60)
    metas = [ChunkMetadata(file_path=f"docs/f{i % 3}.md", folder_path="docs", index_folder="docs", file_name=f"f{i % 3}.md",
                           chunk_index=i // 3, total_chunks=20, start_char=0, end_char=9, indexed_at="2026-01-01T00:00:00",
                           source_modified_at=1_700_000_000 + i, allowed_users=["ann"] if i % 2 else None)
             for i in range(len(texts))]
    ids = vs.store_chunks(list(zip(texts, emb.embed_texts(texts), metas)), sparse_vectors=sp.embed_texts(texts))
    assert vs.delete_by_file("docs/f1.md") == 20
    q = emb.embed_query(texts[0]).tolist()   # plain data: it must outlive the engine it was computed on (the "restart" below)
    sq = tuple(sp.embed_query(texts[0]))
    want = [(c.id, c.score, c.text, c.metadata) for c in vs.search(q, limit=10, sparse_query=sq, sparse_weight=0.2)]
    want_dense = [(c.id, c.score) for c in vs.search(q, limit=5, folder_filter="docs")]
    info = vs.get_collection_info()
    index_dir = str(tmp_path / "persist")
    assert vs.save(index_dir) == index_dir
    # "restart": drop the engine and every host table, point VOITTA_INDEX_DIR at the saved files
    monkeypatch.setenv("VOITTA_INDEX_DIR", index_dir)
    config.get_settings.cache_clear()
    store_registry.reset()
    vector_store._vector_store = None
    vs2 = VectorStoreService()
    assert vs2.get_collection_info() == info
    got = [(c.id, c.score, c.text, c.metadata) for c in vs2.search(q, limit=10, sparse_query=sq, sparse_weight=0.2)]
    assert got == want
    assert [(c.id, c.score) for c in vs2.search(q, limit=5, folder_filter="docs")] == want_dense
    assert vs2.count_by_file("docs/f0.md") == 20 and vs2.count_by_file("docs/f1.md") == 0
    assert set(ids) >= {c[0] for c in got}
    # the restored collection keeps accepting writes (the encoder is re-created too: it lived in the old engine)
    from voitta_rag_amd import embedding, sparse_embedding

    embedding._embedding_service = None
    sparse_embedding._sparse_embedding_service = None
    emb2, sp2 = get_embedding_service(), get_sparse_embedding_service()
    assert np.array_equal(np.asarray(emb2.embed_query(texts[0]), np.float32), np.asarray(q, np.float32))
    more = vs2.store_chunks(list(zip(texts[:5], emb2.embed_texts(texts[:5]), metas[:5])), sparse_vectors=sp2.embed_texts(texts[:5]))
    assert len(more) == 5 and vs2.get_collection_info()["points_count"] == info["points_count"] + 5


def test_compact_after_reindex_keeps_service_answers(native):
    """Re-index = delete_by_file + store_chunks (indexing.py:281-288); compact() reclaims the rows and
    nothing a caller can observe changes."""
    native()
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, VectorStoreService

    rng = np.random.default_rng(21)
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), VectorStoreService()

    def index(fp, texts):
        metas = [ChunkMetadata(file_path=fp, folder_path="docs", index_folder="docs", file_name=os.path.basename(fp),
                               chunk_index=i, total_chunks=len(texts), start_char=0, end_char=9,
                               indexed_at="2026-01-01T00:00:00") for i in range(len(texts))]
        return vs.store_chunks(list(zip(texts, emb.embed_texts(texts), metas)), sparse_vectors=sp.embed_texts(texts))

    t = {fp: _texts(rng, 25) for fp in ("docs/a.md", "docs/b.md", "docs/c.md")}
    for fp, texts in t.items():
        index(fp, texts)
    for _ in range(3):                                   # the same file re-indexed three times
        assert vs.delete_by_file("docs/b.md") == 25
        index("docs/b.md", t["docs/b.md"])
    q, sq = emb.embed_query(t["docs/b.md"][3]), sp.embed_query(t["docs/b.md"][3])
    want = [(c.id, c.score, c.text) for c in vs.search(q, limit=10, sparse_query=sq)]
    info = vs.get_collection_info()
    assert vs.compact(min_dead_fraction=0.9) == 0        # 75 dead of 150: below the threshold
    assert vs.compact() == 75
    assert vs.client.count() == (75, 75)
    assert [(c.id, c.score, c.text) for c in vs.search(q, limit=10, sparse_query=sq)] == want
    assert vs.get_collection_info() == info and vs.count_by_file("docs/b.md") == 25
    assert vs.delete_by_file("docs/a.md") == 25 and vs.compact() == 25 and vs.compact() == 0
    assert [c.metadata.chunk_index for c in vs.get_chunks_by_range("docs/c.md", 3, 5)] == [3, 4, 5]


def test_save_is_atomic_and_a_damaged_snapshot_is_refused_cleanly(native, tmp_path, monkeypatch):
    """A crash or a full disk in the middle of save() must leave the LAST GOOD snapshot loadable (the pointer file
    is the single commit point), and a snapshot whose files do not belong together is refused without leaving a
    half-bound collection behind: the next call retries from scratch."""
    import shutil

    path, shape, w, vocab = native()
    from voitta_rag_amd import config, store_registry, vector_store
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, VectorStoreService

    rng = np.random.default_rng(12)
    emb = get_embedding_service()
    vs = VectorStoreService()
    texts = _texts(rng, 30)
    metas = [ChunkMetadata(file_path=f"docs/f{i % 3}.md", folder_path="docs", index_folder="docs", file_name=f"f{i % 3}.md",
                           chunk_index=i // 3, total_chunks=10, start_char=0, end_char=9, indexed_at="t") for i in range(30)]
    vectors = emb.embed_texts(texts)
    vs.store_chunks(list(zip(texts[:20], vectors[:20], metas[:20])))
    q = vectors[3]
    index_dir = str(tmp_path / "persist")
    vs.save(index_dir)
    want = [(c.id, c.score) for c in vs.search(q, limit=5)]
    first_files = sorted(os.listdir(index_dir))
    assert first_files == ["voitta_documents.g1.payload.jsonl", "voitta_documents.g1.vrindex", "voitta_documents.meta.json"]

    # ---- a save that dies before its commit point (here: while writing the pointer file)
    vs.store_chunks(list(zip(texts[20:], vectors[20:], metas[20:])))
    real = VectorStoreService._write_durably

    def dying(path_, write):
        if path_.endswith(".meta.json"):
            raise OSError(28, "No space left on device")
        return real(path_, write)

    monkeypatch.setattr(VectorStoreService, "_write_durably", staticmethod(dying))
    with pytest.raises(OSError):
        vs.save(index_dir)
    monkeypatch.setattr(VectorStoreService, "_write_durably", staticmethod(real))
    assert set(first_files) <= set(os.listdir(index_dir))  # generation 1 is untouched

    def restart():
        monkeypatch.setenv("VOITTA_INDEX_DIR", index_dir)
        config.get_settings.cache_clear()
        store_registry.reset()
        vector_store._vector_store = None
        return VectorStoreService()

    vs2 = restart()
    assert vs2.get_collection_info()["points_count"] == 20  # the last GOOD snapshot, not a mixture
    assert [(c.id, c.score) for c in vs2.search(q, limit=5)] == want

    # ---- a completed second save replaces generation 1 and removes it
    vs2.store_chunks(list(zip(texts[20:], vectors[20:], metas[20:])))
    vs2.save(index_dir)
    assert sorted(os.listdir(index_dir)) == ["voitta_documents.g2.payload.jsonl", "voitta_documents.g2.vrindex",
                                             "voitta_documents.meta.json"]
    assert restart().get_collection_info()["points_count"] == 30

    # ---- damaged snapshots: refused, nothing half-bound stays behind, a repaired snapshot loads on retry
    payload = os.path.join(index_dir, "voitta_documents.g2.payload.jsonl")
    index = os.path.join(index_dir, "voitta_documents.g2.vrindex")
    shutil.copy(payload, payload + ".keep")
    shutil.copy(index, index + ".keep")
    for damage in ("truncate", "missing", "foreign-index"):
        if damage == "truncate":
            lines = open(payload, encoding="utf-8").read().splitlines(True)
            open(payload, "w", encoding="utf-8").writelines(lines[:10])
        elif damage == "missing":
            os.remove(payload)
        else:
            blob = bytearray(open(index, "rb").read())
            blob[40] ^= 0xFF  # a header field: no longer the file this generation was saved with
            open(index, "wb").write(bytes(blob))
        vs3 = restart()
        with pytest.raises((ValueError, FileNotFoundError, RuntimeError)):
            vs3.search(q, limit=5)
        assert vs3._client is None  # not half-bound ...
        with pytest.raises((ValueError, FileNotFoundError, RuntimeError)):
            vs3.search(q, limit=5)  # ... so a second call fails the same way instead of serving an empty collection
        assert store_registry.get_engine().count() == (0, 0)
        shutil.copy(payload + ".keep", payload)
        shutil.copy(index + ".keep", index)
        assert len(vs3.search(q, limit=5)) == 5  # repaired: the same object recovers without a restart


def test_write_behind_stores_exactly_what_the_three_literal_calls_store(native, monkeypatch):
    """voitta_rag_amd/deferred.py: the per-file sequence of indexing.py:527-560 with embeddings nobody looks at (stored by
    fused vr_index_batch calls from the flusher thread) leaves the same bits in the engine — dense rows, sparse rows,
    filter columns — as the same sequence with VOITTA_DEFERRED_INDEXING=0 (encode -> Python floats -> upsert)."""
    from voitta_rag_amd import deferred, embedding, sparse_embedding, store_registry, vector_store
    from voitta_rag_amd.vector_store import ChunkMetadata

    rng = np.random.default_rng(5)
    files = [(f"d{f % 3}/f{f}.md", _texts(rng, int(rng.integers(1, 40)))) for f in range(25)]
    queries = ["vector database retrieval", "running happily", "kernel memory"]

    def run(defer):
        native(f"wb-model-{int(defer)}")
        monkeypatch.setenv("VOITTA_DEFERRED_INDEXING", "1" if defer else "0")
        emb, sp, vs = embedding.get_embedding_service(), sparse_embedding.get_sparse_embedding_service(), vector_store.get_vector_store()
        for fp, texts in files:
            embeddings = emb.embed_texts(texts)
            sparse_vectors = sp.embed_texts(texts)
            assert isinstance(embeddings, deferred.DeferredEmbeddings) == defer and isinstance(embeddings, list)
            metas = [ChunkMetadata(file_path=fp, folder_path=fp.split("/")[0], index_folder=fp.split("/")[0], file_name=fp.split("/")[1],
                                   chunk_index=i, total_chunks=len(texts), start_char=0, end_char=1, indexed_at="t",
                                   source_modified_at=1_700_000_000 + i) for i in range(len(texts))]
            vs.store_chunks([(t, e, m) for t, e, m in zip(texts, embeddings, metas)], sparse_vectors=sparse_vectors)
            assert not defer or not embeddings.materialized  # nobody looked: the fused path took them
        n = sum(len(t) for _, t in files)
        assert vs.get_collection_info()["points_count"] == n
        engine = vs.client
        dense = engine.get_dense(np.arange(n))
        answers = []
        for q in queries:
            got = vs.search(emb.embed_query(q), limit=10, sparse_query=sp.embed_query(q), sparse_weight=0.3,
                            folder_filter="d1" if q.startswith("k") else None, date_start=1_700_000_002)
            answers.append([(c.metadata.file_path, c.metadata.chunk_index, c.score) for c in got])
            sq = sp.embed_query(q)
            rows, scores = engine.search_sparse(np.array(sq[0], np.int32), np.array(sq[1], np.float32), 50)
            answers.append((rows.tolist(), scores.tolist()))
        stats = engine.stats() if hasattr(engine, "stats") else {}
        return dense, answers, stats

    d1, a1, _ = run(True)
    d0, a0, _ = run(False)
    # The fused calls encode many files' chunks in one batch, the literal calls one file's. The encoder picks its code
    # path by the token count of the batch (large batches: 256-row GEMM tiles and the f16 residual stream; small ones:
    # the short-batch kernels with the f32 residual stream), so the same text may differ by f16-level rounding between
    # the two runs — both inside the 1e-5 |1 - cos| the encoder tests hold against the f64 oracle; the reference's
    # torch path has the same property through padding.
    assert np.max(np.abs(d1 - d0)) < 3e-4 and np.min((d1 * d0).sum(1)) > 1 - 1e-5
    for got, want in zip(a1, a0):
        if isinstance(got, tuple):  # the sparse side is integer-derived: identical
            assert got == want
        else:
            g, w = {x[:2]: x[2] for x in got}, {x[:2]: x[2] for x in want}
            common = set(g) & set(w)
            assert len(common) >= len(w) - 1 and all(abs(g[key] - w[key]) < 2e-4 for key in common)
    store_registry.reset()


def test_a_question_as_text_is_one_engine_call_with_the_same_answer(native):
    """mcp_server.py:469-485 hands embed_query's and the sparse embed_query's results untouched to search(): while
    nobody looks at them they are the question's TEXT and the store answers with one engine call (vr_query_text:
    WordPiece + BM25 tokenise + encode + search). The chunks and scores must be exactly those of the three-call path
    with materialised vectors — hybrid, the dense-only branch (no stem survives), filters, an e5-style prefix, and
    search_many must return the same per query."""
    path, shape, w, vocab = native("e5-mini", "mean")  # "e5" in the name: "query: " / "passage: " prefixes (embedding.py:50,65,82)
    from voitta_rag_amd import deferred
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, get_vector_store

    rng = np.random.default_rng(3)
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    texts = _texts(rng, 300)
    metas = [ChunkMetadata(file_path=f"d{i % 3}/f{i // 10}.md", folder_path=f"d{i % 3}", index_folder=f"d{i % 3}", file_name="f.md",
                           chunk_index=i, total_chunks=300, start_char=0, end_char=1, indexed_at="t",
                           source_modified_at=1_700_000_000 + i) for i in range(300)]
    vs.store_chunks(list(zip(texts, emb.embed_texts(texts), metas)), sparse_vectors=sp.embed_texts(texts))
    key = lambda c: (c.id, c.score, c.text)  # noqa: E731
    questions = ["vector database retrieval", "the of and", "hybrid fusion ranking running?", "x", "kernel memory bandwidth wavefront matrix tile"]
    for qtext in questions:
        for kw in ({}, {"folder_filter": "d1"}, {"exclude_folders": ["d0"], "date_start": 1_700_000_100}, {"limit": 3, "sparse_weight": 0.5}):
            q_ref, s_ref = emb.embed_query(qtext), sp.embed_query(qtext)
            assert isinstance(q_ref, deferred.QueryRef) and not q_ref.materialized and not s_ref.materialized
            got = vs.search(q_ref, sparse_query=s_ref, **kw)                      # the text path
            assert not q_ref.materialized and not s_ref.materialized
            q_arr, s_arr = emb.embed_query(qtext), sp.embed_query(qtext)
            vec = q_arr.tolist()                                                  # looked at: plain floats, three calls
            pair = (list(s_arr[0]), list(s_arr[1]))
            want = vs.search(vec, sparse_query=pair, **kw)
            assert [key(c) for c in got] == [key(c) for c in want], (qtext, kw)
            assert len(got) == min(kw.get("limit", 10), len(got)) and (len(got) > 0 or "folder_filter" in kw or "exclude_folders" in kw)
            # dense only: sparse_query=None
            assert [key(c) for c in vs.search(emb.embed_query(qtext), **kw)] == [key(c) for c in vs.search(vec, **kw)]
    # search_many == search per query (hybrid where the query has stems, dense-only where it has none)
    vecs = np.array([emb.embed_query(q).tolist() for q in questions], np.float32)
    pairs = [tuple(sp.embed_query(q)) for q in questions]
    many = vs.search_many(vecs, limit=7, sparse_queries=pairs, sparse_weight=0.3, exclude_folders=["d2"])
    for i, q in enumerate(questions):
        one = vs.search(vecs[i].tolist(), limit=7, sparse_query=pairs[i], sparse_weight=0.3, exclude_folders=["d2"])
        assert [key(c) for c in many[i]] == [key(c) for c in one], q
