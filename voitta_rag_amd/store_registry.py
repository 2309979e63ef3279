"""Process-wide engine + collection state. The reference keeps all state in the Qdrant server,
so a fresh ``VectorStoreService()`` (api/routes/folders.py:137-143) sees everything; here the
state lives in this registry, never in the service objects (SURVEY.md §8b "Threading")."""
from __future__ import annotations

import threading

from .config import get_settings

_lock = threading.Lock()
_engine = None
_collections: dict = {}


def get_engine():
    """The single Engine of this process (one process per GPU)."""
    global _engine
    with _lock:
        if _engine is None:
            from .engine import Engine

            s = get_settings()
            _engine = Engine(s.embedding_dimension, device=s.gpu, initial_rows=s.initial_rows)
        return _engine


def _stop_collections() -> None:
    """(under _lock) end the write-behind threads of the registered collections before their engine goes away"""
    for col in _collections.values():
        stop = getattr(col, "stop", None)
        if stop is not None:
            stop()


def set_engine(engine) -> None:
    """Tests and multi-GPU launchers install their own engine."""
    global _engine
    with _lock:
        _stop_collections()
        _engine = engine
        _collections.clear()


def collection(name: str, factory):
    with _lock:
        if name not in _collections:
            _collections[name] = factory()
        return _collections[name]


def reset() -> None:
    global _engine
    with _lock:
        _stop_collections()
        if _engine is not None:
            _engine.close()
        _engine = None
        _collections.clear()


def replace(engine, collections: dict) -> None:
    """Install a rebuilt index as the live one (voitta_rag_amd/build_sparse.py with switch=True):
    closes the previous engine, keeps exactly the given {name: collection} host tables."""
    global _engine
    with _lock:
        old = _engine
        _engine = engine
        for col in _collections.values():
            if all(col is not kept for kept in collections.values()) and hasattr(col, "stop"):
                col.stop()
        _collections.clear()
        _collections.update(collections)
    if old is not None and old is not engine:
        old.close()


def forget(name: str) -> None:
    """Drop a host table that was registered only to be saved under its own name."""
    with _lock:
        col = _collections.pop(name, None)
        if col is not None and hasattr(col, "stop"):
            col.stop()
