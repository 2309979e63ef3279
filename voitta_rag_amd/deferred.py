"""Write-behind for the UNMODIFIED caller sequence of the reference.

``IndexingService._index_file_standard`` (src/voitta/services/indexing.py:526-563) does, per file,

    embeddings     = embedder.embed_texts(texts)                     # N x D Python floats in the reference
    sparse_vectors = sparse_embedder.embed_texts(texts)
    chunk_data     = [(chunk.text, embedding, metadata) for chunk, embedding in zip(chunks, embeddings)]
    vector_store.store_chunks(chunk_data, sparse_vectors=sparse_vectors)

— three calls per file, a few dozen chunks each. Run literally on the GPU that is a forward pass of a few
thousand tokens per file (the 256-wide tiles of the GEMMs mostly empty) with N x D floats converted to Python
objects and back in between: 4.3k chunks/s where the engine does 40k.

Here ``embed_texts`` returns a list whose elements are REFERENCES to rows that have not been computed yet
(``DeferredEmbeddings`` / ``EmbeddingRef``), holding the WordPiece ids the host tokenizer already produced;
``SparseEmbeddingService.embed_texts`` does the same with the hashed BM25 stems (``DeferredSparse``). When those
references come back through ``store_chunks`` — the only thing the reference does with them — the store appends the
rows to its host table at once (ids, payloads, counts are immediately right) and queues the token ids; a worker
thread feeds the queue to the engine in batches of thousands of chunks, ONE fused call each (vr_index_batch:
encode -> BM25 tf -> append, nothing leaves HBM), while the caller's thread is already chunking and tokenizing the
next file. Anything that reads the index — search, delete, counts taken from the engine, save, compact — first
waits until every row stored before it is in the engine (read-your-writes, like Qdrant's upsert(wait=True)).

A caller that actually LOOKS at an embedding (indexes it, iterates it, hands it to numpy) gets the numbers: the
reference then triggers the encode of that file's texts on the spot and the list behaves like the plain
``list[list[float]]`` it stands for.

Failure contract (the reference's upsert raises INSIDE store_chunks, and indexing.py:558-590 commits the file's DB
row as soon as the call returns): by default — VOITTA_DEFERRED_INDEXING unset or "sync" — only the REFERENCES are
deferred: ``store_chunks`` itself makes the one fused engine call for its chunks and raises if it fails, exactly
when the reference's call would. The write-behind queue is opt-in, VOITTA_DEFERRED_INDEXING=1, for a caller that
calls ``VectorStoreService.flush()`` (and looks at ``failed_file_paths()``) before it commits its bookkeeping.
VOITTA_DEFERRED_INDEXING=0 turns the whole mechanism off (plain lists of floats)."""
from __future__ import annotations

import os

import numpy as np


def _mode() -> str:
    return os.environ.get("VOITTA_DEFERRED_INDEXING", "sync").strip().lower()


def enabled() -> bool:
    """embed_texts returns references (token ids) instead of floats."""
    return _mode() != "0"


def write_behind() -> bool:
    """store_chunks queues the references for the flusher thread instead of making the fused call itself (opt-in)."""
    return _mode() in ("1", "async", "write-behind")


class EmbeddingRef:
    """Row ``index`` of a DeferredEmbeddings: a sequence of D floats, computed when first looked at."""

    __slots__ = ("batch", "index")

    def __init__(self, batch: "DeferredEmbeddings", index: int):
        self.batch = batch
        self.index = index

    def _row(self) -> np.ndarray:
        return self.batch.array()[self.index]

    def __array__(self, dtype=None, copy=None):
        r = self._row()
        return r if dtype is None else r.astype(dtype)

    def __len__(self) -> int:
        return self.batch.dim

    def __getitem__(self, k):
        v = self._row()[k]
        return float(v) if np.isscalar(v) or getattr(v, "ndim", 1) == 0 else v.tolist()

    def __iter__(self):
        return iter(self._row().tolist())

    def tolist(self) -> list[float]:
        return self._row().tolist()

    def __eq__(self, other):
        try:
            return self.tolist() == list(other)
        except TypeError:
            return NotImplemented

    def __repr__(self) -> str:
        state = "computed" if self.batch.materialized else "deferred"
        return f"<embedding {self.index} of {len(self.batch)} ({state}, dim {self.batch.dim})>"


class DeferredEmbeddings(list):
    """What ``EmbeddingService.embed_texts`` returns: a real ``list`` of EmbeddingRef, plus the token ids the
    rows will be computed from (``ids`` / ``off``: WordPiece ids per text, [CLS] ... [SEP] included)."""

    def __init__(self, encoder, ids: np.ndarray, off: np.ndarray):
        n = int(off.shape[0]) - 1
        super().__init__(EmbeddingRef(self, i) for i in range(n))
        self.encoder = encoder
        self.ids = ids
        self.off = off
        self.dim = int(encoder.desc.hidden)
        self._array: np.ndarray | None = None

    @property
    def materialized(self) -> bool:
        return self._array is not None

    def array(self) -> np.ndarray:
        """The (n, D) f32 embeddings; the forward pass runs on first use."""
        if self._array is None:
            from . import encoder as _enc

            self._array = _enc.encode(self.encoder.engine, self.ids, self.off)
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.array()
        return a if dtype is None else a.astype(dtype)

    def tolist(self) -> list[list[float]]:
        return self.array().tolist()


class QueryEmbedding(list):
    """What ``embed_query`` returns: the plain ``list[float]`` of the reference (embedding.py:76-86), which also keeps
    the f32 array it was made from — ``VectorStoreService.search`` hands that to the engine instead of converting 768
    Python floats back (the two conversions were a tenth of a query's wall time)."""

    def __init__(self, array: np.ndarray):
        super().__init__(array.tolist())
        self.array = np.ascontiguousarray(array, np.float32)


class QueryRef:
    """What ``embed_query`` returns while nobody has looked at it: the QUESTION ITSELF (its text as the encoder will
    see it) standing in for the ``list[float]`` of the reference (embedding.py:76-86). Handed untouched to
    ``VectorStoreService.search`` — all the MCP search tool does with it (mcp_server.py:469-485) — it lets the store
    answer the question in one engine call (vr_query_text: tokenise, encode, search) instead of three calls with
    768 Python floats in between; looked at (indexed, iterated, measured, given to NumPy) it computes the embedding
    and behaves like the list it stands for."""

    def __init__(self, model, text: str):
        self.model = model    # NativeSentenceEncoder
        self.text = text
        self._array: np.ndarray | None = None

    @property
    def materialized(self) -> bool:
        return self._array is not None

    @property
    def array(self) -> np.ndarray:
        if self._array is None:
            self._array = np.ascontiguousarray(self.model.encode(self.text, convert_to_numpy=True), np.float32)
        return self._array

    def __array__(self, dtype=None, copy=None):
        a = self.array
        return a if dtype is None else a.astype(dtype)

    def __len__(self) -> int:
        return int(self.model.desc.hidden)

    def __getitem__(self, k):
        v = self.array[k]
        return float(v) if getattr(v, "ndim", 0) == 0 else v.tolist()

    def __iter__(self):
        return iter(self.array.tolist())

    def tolist(self) -> list[float]:
        return self.array.tolist()

    def __eq__(self, other):
        try:
            return self.tolist() == list(other)
        except TypeError:
            return NotImplemented

    def __repr__(self) -> str:
        return f"<query embedding ({'computed' if self.materialized else 'deferred'}, dim {len(self)})>"


class SparseQueryRef:
    """What the sparse service's ``embed_query`` returns while nobody has looked at it: behaves like the
    ``(indices, values)`` tuple of the reference (sparse_embedding.py:29-39), computed on first use."""

    def __init__(self, text: str, compute):
        self.text = text
        self._compute = compute
        self._pair: tuple[list[int], list[float]] | None = None

    @property
    def materialized(self) -> bool:
        return self._pair is not None

    def _get(self):
        if self._pair is None:
            self._pair = self._compute(self.text)
        return self._pair

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, k):
        return self._get()[k]

    def __len__(self) -> int:
        return 2

    def __bool__(self) -> bool:
        return True

    def __eq__(self, other):
        try:
            return self._get() == tuple(other)
        except TypeError:
            return NotImplemented

    def __repr__(self) -> str:
        return f"<bm25 query vector ({'computed' if self.materialized else 'deferred'})>"


class SparseRef:
    """Entry ``index`` of a DeferredSparse: behaves like the (indices, values) tuple it stands for."""

    __slots__ = ("batch", "index")

    def __init__(self, batch: "DeferredSparse", index: int):
        self.batch = batch
        self.index = index

    def _pair(self) -> tuple[list[int], list[float]]:
        return self.batch.rows()[self.index]

    def __iter__(self):
        return iter(self._pair())

    def __getitem__(self, k):
        return self._pair()[k]

    def __len__(self) -> int:
        return 2

    def __eq__(self, other):
        try:
            return self._pair() == tuple(other)
        except TypeError:
            return NotImplemented

    def __repr__(self) -> str:
        return f"<bm25 vector {self.index} of {len(self.batch)}>"


class DeferredSparse(list):
    """What ``SparseEmbeddingService.embed_texts`` returns: a real ``list`` of SparseRef, plus the hashed stems
    (``stems`` / ``off``) the term weights will be computed from."""

    def __init__(self, engine, off: np.ndarray, stems: np.ndarray):
        n = int(off.shape[0]) - 1
        super().__init__(SparseRef(self, i) for i in range(n))
        self.engine = engine
        self.off = off
        self.stems = stems
        self._rows: list | None = None

    @property
    def materialized(self) -> bool:
        return self._rows is not None

    def rows(self) -> list[tuple[list[int], list[float]]]:
        if self._rows is None:
            self._rows = [(idx.tolist(), val.tolist()) for idx, val in self.engine.bm25_tf(self.off, self.stems)]
        return self._rows


def take_deferred(chunks, sparse_vectors):
    """If every embedding of ``chunks`` is an untouched EmbeddingRef of ONE DeferredEmbeddings and ``sparse_vectors`` is
    None or an untouched DeferredSparse covering every chunk, return (wp_ids, wp_off, bm_ids, bm_off) for exactly
    these chunks in their order; else None (the caller then takes the ordinary path, which computes them)."""
    n = len(chunks)
    first = chunks[0][1]
    if not isinstance(first, EmbeddingRef):
        return None
    batch = first.batch
    if batch.materialized:
        return None
    idx = np.empty(n, np.int64)
    for i, c in enumerate(chunks):
        ref = c[1]
        if not isinstance(ref, EmbeddingRef) or ref.batch is not batch:
            return None
        idx[i] = ref.index
    whole = n == len(batch) and np.array_equal(idx, np.arange(n))
    if whole:
        wp_ids, wp_off = batch.ids, batch.off
    else:
        lens = (batch.off[idx + 1] - batch.off[idx]).astype(np.int64)
        wp_off = np.zeros(n + 1, np.int32)
        wp_off[1:] = np.cumsum(lens)
        wp_ids = np.concatenate([batch.ids[batch.off[j]:batch.off[j + 1]] for j in idx]) if n else np.zeros(0, np.int32)
    bm_ids = bm_off = None
    if sparse_vectors is not None and len(sparse_vectors) > 0:
        if not isinstance(sparse_vectors, DeferredSparse) or sparse_vectors.materialized or len(sparse_vectors) < n:
            return None
        if len(sparse_vectors) == n:
            bm_off, bm_ids = sparse_vectors.off, sparse_vectors.stems
        else:  # vector i belongs to chunk i (vector_store.py:291): the leading n of them
            bm_off = sparse_vectors.off[:n + 1]
            bm_ids = sparse_vectors.stems[:int(bm_off[-1])]
    return wp_ids, np.ascontiguousarray(wp_off, np.int32), bm_ids, bm_off
