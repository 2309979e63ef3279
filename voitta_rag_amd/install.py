"""Switches a voitta-rag process to the native services without touching its code:

    import voitta_rag_amd.install            # before voitta.services is first used
    voitta_rag_amd.install.install()

replaces the three module-level singletons the callers obtain:
``voitta.services.embedding.get_embedding_service`` (embedding.py:93-98),
``voitta.services.sparse_embedding.get_sparse_embedding_service`` (sparse_embedding.py:57-62) and
``voitta.services.vector_store.get_vector_store`` (vector_store.py:1023-1028), plus the classes the
callers construct directly (api/routes/folders.py:137-143 builds ``VectorStoreService()``).
IndexingService (indexing.py:180-191) and the MCP search tool (mcp_server.py:413-414,469-485) then
run unmodified on the MI355X engine."""
from __future__ import annotations

import importlib
import sys
import types


def install(chunking: bool = True) -> None:
    """chunking=True also swaps ChunkingService / get_chunking_service (services/chunking.py:19-246,
    obtained by IndexingService at indexing.py:187) for the native chunker."""
    from . import chunking as native_chunking
    from . import embedding, sparse_embedding, vector_store

    mapping = {
        "voitta.services.embedding": (embedding, ["EmbeddingService", "get_embedding_service"]),
        "voitta.services.sparse_embedding": (sparse_embedding, ["SparseEmbeddingService", "get_sparse_embedding_service",
                                                                "SPARSE_VECTOR_NAME"]),
        "voitta.services.vector_store": (vector_store, ["VectorStoreService", "get_vector_store", "ChunkMetadata",
                                                        "StoredChunk"]),
    }
    if chunking:
        mapping["voitta.services.chunking"] = (native_chunking, ["ChunkingService", "get_chunking_service", "Chunk"])
    for name, (native, attrs) in mapping.items():
        try:
            mod = importlib.import_module(name)
        except Exception:
            # the reference module cannot import here (e.g. qdrant_client missing): provide it whole
            mod = types.ModuleType(name)
            sys.modules[name] = mod
        for a in attrs:
            setattr(mod, a, getattr(native, a))
    pkg = sys.modules.get("voitta.services")
    if pkg is not None:  # names re-exported by services/__init__.py:3-26
        for _, (native, attrs) in mapping.items():
            for a in attrs:
                if hasattr(pkg, a):
                    setattr(pkg, a, getattr(native, a))
