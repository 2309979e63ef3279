"""Drop-in for the reference's SparseEmbeddingService (src/voitta/services/sparse_embedding.py):
same constant, class, method names, argument meaning and return shapes. fastembed's
``SparseTextEmbedding("Qdrant/bm25")`` is replaced by the C++ text pipeline (vr_bm25_tokenize) and
the HIP token-count / tf kernel (vr_bm25_tf)."""
from __future__ import annotations

import logging

import numpy as np

from . import bm25 as _bm25
from .store_registry import get_engine

logger = logging.getLogger(__name__)

SPARSE_VECTOR_NAME = "bm25"  # sparse_embedding.py:9


class SparseEmbeddingService:
    """Service for generating BM25 sparse embeddings on the GPU."""

    def __init__(self):
        self._model = None

    @property
    def model(self):
        """The engine that carries the BM25 kernels (lazy, like sparse_embedding.py:18-27)."""
        if self._model is None:
            logger.info("Binding BM25 sparse embedding to the native engine")
            self._model = get_engine()
        return self._model

    def embed_query(self, query: str) -> tuple[list[int], list[float]]:
        """(indices, values): the set of token ids, every value 1.0 (Bm25.query_embed);
        ``([], [])`` when nothing survives stop-word removal (sparse_embedding.py:36-37)."""
        _, ids = _bm25.hashed_stems([query])
        if ids.size == 0:
            return [], []
        uniq = np.unique(ids)
        return uniq.tolist(), [1.0] * int(uniq.size)

    def embed_texts(self, texts: list[str]) -> list[tuple[list[int], list[float]]]:
        """List of (indices, values); values are Python floats (f64) as fastembed returns them.
        Indices come out ascending (fastembed: first-occurrence order; a sparse vector is a set)."""
        if not texts:
            return []
        off, ids = _bm25.hashed_stems(texts)
        rows = self.model.bm25_tf(off, ids)
        return [(i.tolist(), v.tolist()) for i, v in rows]


_sparse_embedding_service: SparseEmbeddingService | None = None


def get_sparse_embedding_service() -> SparseEmbeddingService:
    global _sparse_embedding_service
    if _sparse_embedding_service is None:
        _sparse_embedding_service = SparseEmbeddingService()
    return _sparse_embedding_service
