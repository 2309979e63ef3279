"""Host logic of BulkIndexer (voitta_rag_amd/indexer.py) without a GPU: the native chunker and BM25
tokeniser run on the CPU, the encoder and the store are stand-ins that record what they are handed.
Checks the cut into batches across file boundaries, the per-file counts, metadata, e5 prefixes and
error propagation from the producer thread."""
import numpy as np
import pytest

from oracle import chunking as ochunk
from voitta_rag_amd.chunking import ChunkingService
from voitta_rag_amd.indexer import BulkIndexer, ParsedFile


class _Model:
    def __init__(self):
        self.seen = []

    def tokenize(self, texts):
        self.seen.extend(texts)
        lens = [len(t.split()) + 2 for t in texts]
        off = np.zeros(len(texts) + 1, np.int32)
        off[1:] = np.cumsum(lens)
        return np.zeros(int(off[-1]), np.int32), off


class _Embedder:
    def __init__(self, name):
        self.model_name = name
        self.model = _Model()


class _Store:
    def __init__(self, fail_at=None):
        self.calls = []
        self.fail_at = fail_at

    def index_chunks(self, texts, metadatas, wp_ids, wp_off, bm_ids=None, bm_off=None):
        if self.fail_at is not None and len(self.calls) == self.fail_at:
            raise RuntimeError("store is full")
        assert len(texts) == len(metadatas) == len(wp_off) - 1
        if bm_off is not None:
            assert len(bm_off) == len(texts) + 1 and bm_off[-1] == len(bm_ids)
        self.calls.append((list(texts), list(metadatas)))
        return [str(i) for i in range(len(texts))]


def _files(n, rng):
    words = "vector index query chunk dense sparse fusion kernel memory tile".split()
    out = []
    for i in range(n):
        paras = [" ".join(rng.choice(words, size=int(rng.integers(8, 40)))) + "." for _ in range(int(rng.integers(1, 9)))]
        content = "" if i == 3 else "\n\n".join(paras)
        out.append(ParsedFile(content, f"d{i % 3}/f{i}.md", f"d{i % 3}", f"d{i % 3}", f"f{i}.md",
                              source_modified_at=1000 + i, allowed_users=["u"] if i == 1 else None))
    return out


@pytest.mark.parametrize("batch_chunks,files_per_cut", [(7, 2), (64, 5), (100000, 64)])
def test_batches_cut_across_files_and_counts(batch_chunks, files_per_cut):
    rng = np.random.default_rng(4)
    files = _files(23, rng)
    chunker = ChunkingService(90, 10, "recursive")
    store, emb = _Store(), _Embedder("bge-small")
    counts = BulkIndexer(chunker, emb, store, sparse=True, batch_chunks=batch_chunks,
                         files_per_cut=files_per_cut).index_files(iter(files))
    want = {f.file_path: len(ochunk.chunk_text(f.content, 90, 10, "recursive")) for f in files}
    assert counts == want and want["d0/f3.md"] == 0
    texts = [t for call in store.calls for t in call[0]]
    metas = [m for call in store.calls for m in call[1]]
    assert all(len(call[0]) <= batch_chunks for call in store.calls)
    assert len(store.calls) == -(-sum(want.values()) // batch_chunks)  # batches are full except the last
    expect = [(c[0], f.file_path, c[1], c[2], c[3]) for f in files for c in ochunk.chunk_text(f.content, 90, 10, "recursive")]
    assert [(t, m.file_path, m.chunk_index, m.start_char, m.end_char) for t, m in zip(texts, metas)] == expect
    assert all(m.total_chunks == want[m.file_path] and m.folder_path == m.index_folder for m in metas)
    assert {m.allowed_users[0] for m in metas if m.allowed_users} == {"u"}
    assert emb.model.seen == texts  # no prefix for a non-e5 model


def test_e5_prefix_goes_to_the_encoder_only():
    files = _files(4, np.random.default_rng(5))
    store, emb = _Store(), _Embedder("intfloat/e5-base-v2")
    BulkIndexer(ChunkingService(90, 10, "recursive"), emb, store, sparse=False, batch_chunks=16).index_files(files)
    stored = [t for call in store.calls for t in call[0]]
    assert emb.model.seen == ["passage: " + t for t in stored]  # embedding.py:65-66; the stored text stays bare


def test_errors_of_either_thread_surface_and_do_not_hang():
    files = _files(30, np.random.default_rng(6))
    with pytest.raises(RuntimeError, match="store is full"):
        BulkIndexer(ChunkingService(90, 10, "recursive"), _Embedder("m"), _Store(fail_at=1), batch_chunks=5).index_files(files)

    def broken():
        yield files[0]
        raise OSError("parser died")

    with pytest.raises(OSError, match="parser died"):
        BulkIndexer(ChunkingService(90, 10, "recursive"), _Embedder("m"), _Store(), batch_chunks=5).index_files(broken())
