"""Bulk front-end of the indexing path: parsed files in, stored rows out.

The reference indexes one file at a time (IndexingService._index_file_standard,
src/voitta/services/indexing.py:513-563): ``chunker.chunk_text`` → ``embedder.embed_texts`` →
``sparse_embedder.embed_texts`` → ``vector_store.store_chunks``, each step waiting for the one
before and N×D Python floats passing between them. ``BulkIndexer`` runs the same four steps for a
stream of files with the same stored result (same chunks, payload fields, vectors and scores), but

* cut into batches of ``batch_chunks`` chunks regardless of file boundaries (a 3-chunk file does not
  pay a GPU launch sequence of its own),
* the host stages — chunking (vr_chunk_texts), WordPiece (vr_wordpiece_encode) and the BM25 tokeniser
  (vr_bm25_tokenize), all on every host thread — run on a producer thread for batch i+1 while
* point ids and payload dicts of batch i+1 are made there too (VectorStoreService.prepare_rows), and
* the GPU stages of batch i run as ONE engine call (vr_index_batch: encode → tf → append, nothing
  leaves HBM) on the caller's thread. ctypes drops the GIL inside both, so they overlap.

What the caller owns stays with the caller: parsing, content hashes, the SQL ``IndexedFile`` rows
(indexing.py:565-600). ``index_files`` returns the chunk count per file for that bookkeeping."""
from __future__ import annotations

import queue
import threading
from dataclasses import dataclass
from datetime import datetime, timezone
from typing import Iterable, Iterator

from . import bm25 as _bm25
from .chunking import ChunkingService, get_chunking_service
from .embedding import EmbeddingService, get_embedding_service
from .vector_store import ChunkMetadata, VectorStoreService, get_vector_store


@dataclass
class ParsedFile:
    """What _index_file_standard holds once a file is parsed (indexing.py:486-512,537-552)."""

    content: str
    file_path: str
    folder_path: str
    index_folder: str
    file_name: str
    source_created_at: int | None = None
    source_modified_at: int | None = None
    allowed_users: list[str] | None = None
    source_url: str | None = None


@dataclass
class _Batch:
    texts: list[str]
    metadatas: list[ChunkMetadata]
    wp_ids: object
    wp_off: object
    bm_ids: object
    bm_off: object
    counts: dict[str, int]
    rows: tuple | None = None  # (point ids, payload dicts) made on the producer thread


class BulkIndexer:
    def __init__(self, chunker: ChunkingService | None = None, embedder: EmbeddingService | None = None,
                 vector_store: VectorStoreService | None = None, sparse: bool = True, batch_chunks: int = 4096,
                 files_per_cut: int = 64):
        self.chunker = chunker or get_chunking_service()
        self.embedder = embedder or get_embedding_service()
        self.vector_store = vector_store or get_vector_store()
        self.sparse = sparse
        self.batch_chunks = int(batch_chunks)
        self.files_per_cut = int(files_per_cut)

    # ---- host stages (producer thread) -------------------------------------------------------------
    def _tokenise(self, texts: list[str], metadatas: list[ChunkMetadata], counts: dict[str, int]) -> _Batch:
        model = self.embedder.model
        encoder_texts = texts
        if "e5" in self.embedder.model_name.lower():  # embedding.py:65-66
            encoder_texts = [f"passage: {t}" for t in texts]
        wp_ids, wp_off = model.tokenize(encoder_texts)
        bm_ids = bm_off = None
        if self.sparse:
            bm_off, bm_ids = _bm25.hashed_stems(texts)  # BM25 sees the chunk text itself (indexing.py:529-530)
        rows = self.vector_store.prepare_rows(texts, metadatas) if hasattr(self.vector_store, "prepare_rows") else None
        return _Batch(texts, metadatas, wp_ids, wp_off, bm_ids, bm_off, counts, rows)

    def _batches(self, files: Iterable[ParsedFile]) -> Iterator[_Batch]:
        texts: list[str] = []
        metadatas: list[ChunkMetadata] = []
        counts: dict[str, int] = {}
        pending: list[ParsedFile] = []

        def cut(group: list[ParsedFile]):
            indexed_at = datetime.now(timezone.utc).isoformat()  # indexing.py:538
            for f, chunks in zip(group, self.chunker.chunk_texts([f.content for f in group])):
                counts[f.file_path] = len(chunks)  # 0: empty content / no chunks (indexing.py:509-522)
                for c in chunks:
                    texts.append(c.text)
                    metadatas.append(ChunkMetadata(
                        file_path=f.file_path, folder_path=f.folder_path, index_folder=f.index_folder,
                        file_name=f.file_name, chunk_index=c.index, total_chunks=len(chunks),
                        start_char=c.start_char, end_char=c.end_char, indexed_at=indexed_at,
                        source_created_at=f.source_created_at, source_modified_at=f.source_modified_at,
                        allowed_users=f.allowed_users, source_url=f.source_url))

        def drain(everything: bool):
            nonlocal texts, metadatas, counts
            while len(texts) >= self.batch_chunks or (everything and texts):
                take = min(len(texts), self.batch_chunks)
                last = take == len(texts)
                yield self._tokenise(texts[:take], metadatas[:take], counts if last else {})
                texts, metadatas = texts[take:], metadatas[take:]
                if last:
                    counts = {}

        for f in files:
            pending.append(f)
            if len(pending) >= self.files_per_cut:
                cut(pending)
                pending = []
                yield from drain(False)
        if pending:
            cut(pending)
        yield from drain(True)
        if counts:  # files that produced no chunk at all after the last batch
            yield _Batch([], [], None, None, None, None, counts, None)

    # ---- the pipeline --------------------------------------------------------------------------------
    def index_files(self, files: Iterable[ParsedFile]) -> dict[str, int]:
        """Index every file; returns {file_path: chunk_count} (0 = nothing to index, the reference's
        ``return False, 0`` cases at indexing.py:509-522). A file's count is reported with the batch that
        holds its last chunk."""
        self.embedder.model  # noqa: B018  load the encoder before the threads start
        q: queue.Queue = queue.Queue(maxsize=2)
        failure: list[BaseException] = []

        def produce():
            try:
                for batch in self._batches(files):
                    q.put(batch)
            except BaseException as exc:  # surfaced on the caller's thread
                failure.append(exc)
            finally:
                q.put(None)

        producer = threading.Thread(target=produce, name="voitta-tokenise", daemon=True)
        producer.start()
        done: dict[str, int] = {}
        try:
            while True:
                batch = q.get()
                if batch is None:
                    break
                if batch.texts:
                    if batch.rows is not None:
                        self.vector_store.index_chunks(batch.texts, batch.metadatas, batch.wp_ids, batch.wp_off,
                                                       batch.bm_ids, batch.bm_off, rows=batch.rows)
                    else:
                        self.vector_store.index_chunks(batch.texts, batch.metadatas, batch.wp_ids, batch.wp_off,
                                                       batch.bm_ids, batch.bm_off)
                done.update(batch.counts)
        finally:
            while producer.is_alive():  # unblock a producer stuck on a full queue after an error here
                try:
                    q.get_nowait()
                except queue.Empty:
                    producer.join(0.01)
        if failure:
            raise failure[0]
        return done
