"""MI355X-native indexing-and-retrieval core for voitta-rag's hot path.

Drop-in service classes (same names, signatures and return types as the reference's
``voitta.services``):

  EmbeddingService        <- src/voitta/services/embedding.py
  SparseEmbeddingService  <- src/voitta/services/sparse_embedding.py
  VectorStoreService      <- src/voitta/services/vector_store.py

all driving ``libvoitta_engine.so`` (hand-written gfx950 HIP kernels behind the C-ABI of
``include/voitta_engine.h``) through ctypes. There is no CPU fallback: importing the service
classes works anywhere, but creating an engine without the built library and a gfx950 device
raises.
"""

from ._lib import EngineError, library_path, load_library  # noqa: F401
from .engine import Engine, SearchFilter  # noqa: F401

__all__ = ["Engine", "SearchFilter", "EngineError", "load_library", "library_path"]
