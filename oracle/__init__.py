"""TEST INFRASTRUCTURE — CPU oracles for the voitta-rag hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package. The product (``voitta_rag_amd``) never does: it has no CPU fallback.

Modules
  core      ctypes wrapper over oracle_core.c (cosine preprocessing, exact f32 dense scores,
            Qdrant-IDF sparse scores, ranking)
  fusion    line-by-line Python restatement of VectorStoreService._hybrid_search
  filters   restatement of VectorStoreService._build_filter over payload dicts
  bm25      restatement of fastembed's Qdrant/bm25 text pipeline
  bert      NumPy restatement of the sentence-transformers encode pipeline (BERT encoder,
            pooling, normalisation), pinned against transformers.BertModel in this container

PARITY UNPINNED: the reference's own tests hold no fixture for any of these computations
(SURVEY.md F6); every [EXT] behaviour is pinned only by the hand-derived known answers under
tests/golden/ and, for bert, by transformers' BertModel run here with seeded weights.
"""
