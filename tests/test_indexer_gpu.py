"""BulkIndexer (voitta_rag_amd/indexer.py): the pipelined, cross-file batched form of
IndexingService._index_file_standard's chunk → embed → sparse embed → store sequence
(src/voitta/services/indexing.py:513-563) must leave the store in the state the sequence itself
leaves it in: same chunks and payload fields, same BM25 rows (bit-exact), the same dense vectors up to
the encoder's batch-shape rounding (a 3-row batch takes the skinny split-K GEMM, a 200-row batch the
256-tile kernel: different summation orders, |1 - cos| ~1e-7), and therefore the same answers."""
import numpy as np
import pytest

from test_services_gpu import WORDS, native  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _document(rng, paragraphs):
    paras = []
    for _ in range(paragraphs):
        sents = [" ".join(rng.choice(WORDS, size=int(rng.integers(3, 14)))) + rng.choice([".", "!", "?"])
                 for _ in range(int(rng.integers(1, 6)))]
        paras.append(" ".join(sents))
    return "\n\n".join(paras)


def _files(rng):
    from voitta_rag_amd.indexer import ParsedFile

    spec = [("docs/a.md", "docs", "docs", 9), ("docs/sub/b.md", "docs/sub", "docs", 1), ("notes/c.txt", "notes", "notes", 14),
            ("notes/empty.txt", "notes", "notes", 0), ("d.txt", "", "", 4), ("blank.md", "", "", -1),
            ("docs/e.md", "docs", "docs", 22), ("docs/f.md", "docs", "docs", 2)]
    files = []
    for i, (fp, folder, index_folder, paragraphs) in enumerate(spec):
        content = "" if paragraphs == 0 else " \n\t " if paragraphs < 0 else _document(rng, paragraphs)
        files.append(ParsedFile(content=content, file_path=fp, folder_path=folder, index_folder=index_folder,
                                file_name=fp.rsplit("/", 1)[-1], source_created_at=1_700_000_000 + i if i % 2 else None,
                                source_modified_at=1_710_000_000 + 10 * i, allowed_users=["ann"] if i == 2 else None,
                                source_url=f"https://example.test/{i}" if i == 4 else None))
    return files


QUERIES = ["vector database index", "running happily", "memory bandwidth of the matrix kernel", "hybrid fusion ranking?"]


def _snapshot(vs, emb, sp):
    """Everything observable about the store: payloads in row order, stored vectors, answers."""
    col = vs._col
    payloads = [{k: v for k, v in p.items() if k != "indexed_at"} for p in col.payload]
    dense = vs.client.get_dense(np.arange(len(payloads)))
    answers = []
    for q in QUERIES:
        qv = emb.embed_query(q)
        for kw in ({}, {"sparse_query": sp.embed_query(q), "sparse_weight": 0.3}, {"include_folders": ["docs"]},
                   {"sparse_query": sp.embed_query(q), "sparse_weight": 1.0}):
            res = vs.search(qv, limit=6, **kw)
            answers.append([((r.metadata.file_path, r.metadata.chunk_index), r.score) for r in res])
    return payloads, dense, answers


def test_bulk_indexer_matches_the_per_file_sequence(native, monkeypatch):  # noqa: F811
    monkeypatch.setenv("CHUNK_SIZE", "120")
    monkeypatch.setenv("CHUNK_OVERLAP", "20")
    rng = np.random.default_rng(11)
    native()
    from voitta_rag_amd.chunking import get_chunking_service
    from voitta_rag_amd.embedding import get_embedding_service
    from voitta_rag_amd.indexer import BulkIndexer
    from voitta_rag_amd.sparse_embedding import get_sparse_embedding_service
    from voitta_rag_amd.vector_store import ChunkMetadata, get_vector_store

    files = _files(rng)

    # A: the reference's sequence, one file at a time (indexing.py:513-563)
    chunker, emb, sp, vs = get_chunking_service(), get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    assert (chunker.chunk_size, chunker.chunk_overlap) == (120, 20)
    want_counts = {}
    for f in files:
        chunks = chunker.chunk_text(f.content) if f.content.strip() else []
        want_counts[f.file_path] = len(chunks)
        if not chunks:
            continue
        texts = [c.text for c in chunks]
        embeddings, sparse_vectors = emb.embed_texts(texts), sp.embed_texts(texts)
        vs.store_chunks([(c.text, e, ChunkMetadata(
            file_path=f.file_path, folder_path=f.folder_path, index_folder=f.index_folder, file_name=f.file_name,
            chunk_index=c.index, total_chunks=len(chunks), start_char=c.start_char, end_char=c.end_char, indexed_at="t",
            source_created_at=f.source_created_at, source_modified_at=f.source_modified_at,
            allowed_users=f.allowed_users, source_url=f.source_url)) for c, e in zip(chunks, embeddings)],
            sparse_vectors=sparse_vectors)
    assert sum(want_counts.values()) > 60 and want_counts["notes/empty.txt"] == 0 and want_counts["blank.md"] == 0
    a_payloads, a_dense, a_answers = _snapshot(vs, emb, sp)

    # B: a fresh engine and store, the pipelined bulk form with batches that cut across files
    native("mini-model-b")
    emb, sp, vs = get_embedding_service(), get_sparse_embedding_service(), get_vector_store()
    counts = BulkIndexer(batch_chunks=16, files_per_cut=3).index_files(iter(files))
    assert counts == want_counts
    b_payloads, b_dense, b_answers = _snapshot(vs, emb, sp)

    assert b_payloads == a_payloads
    cos = (a_dense * b_dense).sum(1)
    assert a_dense.shape == b_dense.shape and np.max(np.abs(1 - cos)) < 1e-5
    assert len(a_answers) == len(b_answers)
    for got, want in zip(b_answers, a_answers):
        assert len(got) == len(want)
        # min-max fusion divides by the spread of the prefetched scores (a few 1e-2 here), so the 1e-6
        # differences between the two GEMM paths' dense scores come back a hundredfold in the fused ones
        assert np.allclose([s for _, s in got], [s for _, s in want], atol=1e-3)
        if [k for k, _ in got] != [k for k, _ in want]:  # only near-ties may swap
            assert sorted(k for k, _ in got[:-1]) == sorted(k for k, _ in want[:-1]) or \
                set(k for k, _ in got) == set(k for k, _ in want)
    # sparse-only answers do not depend on the encoder at all: bit-exact
    for i in range(3, len(a_answers), 4):
        assert b_answers[i] == a_answers[i]


def test_bulk_indexer_surfaces_producer_errors(native):  # noqa: F811
    native()
    from voitta_rag_amd.indexer import BulkIndexer, ParsedFile

    def files():
        yield ParsedFile("vector database index. " * 30, "a.md", "", "", "a.md")
        raise OSError("parser died")

    with pytest.raises(OSError, match="parser died"):
        BulkIndexer(batch_chunks=8).index_files(files())
