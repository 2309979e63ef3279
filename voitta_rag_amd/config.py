"""Settings of the native services. Same environment variables and defaults as the reference's
``Settings`` for the hot path (src/voitta/config.py:28-44,72), plus the few the native engine adds
(SURVEY.md §5 "Config / flags"). EMBEDDING_MODEL must be a local checkpoint directory: there is no
network, so hub names cannot be resolved."""
from __future__ import annotations

import os
from functools import lru_cache


class Settings:
    def __init__(self):
        self.qdrant_collection: str = os.getenv("QDRANT_COLLECTION", "voitta_documents")  # config.py:30
        self.embedding_model: str = os.getenv("EMBEDDING_MODEL", "intfloat/e5-base-v2")    # config.py:33
        self.embedding_dimension: int = int(os.getenv("EMBEDDING_DIMENSION", "768"))       # config.py:34
        self.embedding_device: str = os.getenv("EMBEDDING_DEVICE", "auto")                 # config.py:36
        self.chunk_size: int = int(os.getenv("CHUNK_SIZE", "512"))                         # config.py:39
        self.chunk_overlap: int = int(os.getenv("CHUNK_OVERLAP", "50"))                    # config.py:40
        self.chunking_strategy: str = os.getenv("CHUNKING_STRATEGY", "recursive")          # config.py:41
        self.sparse_weight: float = float(os.getenv("SPARSE_WEIGHT", "0.1"))               # config.py:44
        self.mcp_search_limit: int = int(os.getenv("MCP_SEARCH_LIMIT", "20"))              # config.py:72
        # native additions
        self.gpu: int = int(os.getenv("VOITTA_GPU", os.getenv("LOCAL_RANK", "0")))
        self.initial_rows: int = int(os.getenv("VOITTA_INITIAL_ROWS", "0"))
        # directory of the persisted index (VectorStoreService.save / load); loaded on first use when present
        self.index_dir: str = os.getenv("VOITTA_INDEX_DIR", "")


@lru_cache
def get_settings() -> Settings:
    return Settings()
