"""Drop-in for the reference's SparseEmbeddingService (src/voitta/services/sparse_embedding.py):
same constant, class, method names, argument meaning and return shapes. fastembed's
``SparseTextEmbedding("Qdrant/bm25")`` is replaced by the C++ text pipeline (vr_bm25_tokenize) and
the HIP token-count / tf kernel (vr_bm25_tf)."""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np

from . import bm25 as _bm25
from . import deferred as _deferred
from .store_registry import get_engine

SPARSE_VECTOR_NAME = "bm25"  # the named sparse vector of the collection (sparse_embedding.py:9)
_log = logging.getLogger(__name__)


def _query_vector(query: str) -> tuple[list[int], list[float]]:
    """Bm25.query_embed: the SET of hashed stems, every value 1.0; nothing survives
    stop-word removal -> ``([], [])`` (sparse_embedding.py:36-37)."""
    stems = _bm25.hashed_stems([query])[1]
    distinct = np.unique(stems)
    return (distinct.tolist(), [1.0] * int(distinct.size)) if stems.size else ([], [])


class SparseEmbeddingService:
    """BM25 sparse embeddings; term weights are computed on the GPU, the text side on the host."""

    def __init__(self) -> None:
        self._model = None  # the engine that carries the BM25 kernels, bound on first use (:18-27)

    def embed_texts(self, texts: list[str]) -> list[tuple[list[int], list[float]]]:
        """One (indices, values) pair per text; values are Python floats (f64) as fastembed returns
        them. Indices come out ascending (fastembed: first-occurrence order; a sparse vector is a set)."""
        if len(texts) == 0:
            return []
        offsets, stems = _bm25.hashed_stems(texts)
        if _deferred.enabled():  # the tf weights are computed when looked at, or inside the fused store
            return _deferred.DeferredSparse(self.model, offsets, stems)
        return [(idx.tolist(), val.tolist()) for idx, val in self.model.bm25_tf(offsets, stems)]

    def embed_query(self, query: str) -> tuple[list[int], list[float]]:
        if _deferred.enabled():
            return _deferred.SparseQueryRef(query, _query_vector)  # computed when looked at, or inside the search call
        return _query_vector(query)

    @property
    def model(self):
        engine = self._model
        if engine is None:
            _log.info("Binding BM25 sparse embedding to the native engine")
            engine = self._model = get_engine()
        return engine


_sparse_embedding_service: Optional[SparseEmbeddingService] = None


def get_sparse_embedding_service() -> SparseEmbeddingService:
    """The process-wide instance (sparse_embedding.py:57-62)."""
    global _sparse_embedding_service
    service = _sparse_embedding_service
    if service is None:
        service = _sparse_embedding_service = SparseEmbeddingService()
    return service
