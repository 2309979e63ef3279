"""Drop-in for the reference's ``ChunkingService`` (src/voitta/services/chunking.py:9-246), the step
in front of the indexing path (services/indexing.py:380,515): same class, same
``Chunk`` fields, same constructor defaults (``x or settings.x``, so 0 falls back to the setting just
as it does there). The splitting runs in the native library (vr_chunk_texts, csrc/chunking.cpp), one
document per host thread; ``chunk_texts`` is the batched form the indexer uses to cut a whole folder
at once. Parity unpinned: see csrc/chunking.cpp."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import check, load_library
from .config import get_settings

_STRATEGY = {"recursive": 0, "sentence": 1, "fixed": 2}


@dataclass
class Chunk:
    """A chunk of text with its position (reference: chunking.py:9-16)."""

    text: str
    index: int
    start_char: int
    end_char: int


class ChunkingService:
    def __init__(self, chunk_size: int | None = None, chunk_overlap: int | None = None, strategy: str | None = None):
        settings = get_settings()
        self.chunk_size = chunk_size or settings.chunk_size            # chunking.py:29
        self.chunk_overlap = chunk_overlap or settings.chunk_overlap   # chunking.py:30
        self.strategy = strategy or settings.chunking_strategy         # chunking.py:31
        self._lib = load_library()

    def chunk_text(self, text: str) -> list[Chunk]:
        """chunking.py:33-45."""
        return self.chunk_texts([text])[0]

    def chunk_texts(self, texts: list[str]) -> list[list[Chunk]]:
        n = len(texts)
        if n == 0:
            return []
        raw = [(t or "").encode("utf-8", "surrogatepass") for t in texts]
        arr = (C.c_char_p * n)(*raw)
        lens = np.asarray([len(b) for b in raw], np.int64)
        handle = C.c_void_p()
        check(self._lib.vr_chunk_texts(arr, lens.ctypes.data_as(C.POINTER(C.c_int64)), n, int(self.chunk_size),
                                       int(self.chunk_overlap), _STRATEGY.get(self.strategy, 0), C.byref(handle)))
        try:
            count = C.c_int64()
            doc_off, span, text_off = (C.POINTER(C.c_int64)() for _ in range(3))
            blob = C.POINTER(C.c_char)()
            check(self._lib.vr_chunks_view(handle, C.byref(count), C.byref(doc_off), C.byref(span), C.byref(text_off),
                                           C.byref(blob)))
            m = count.value
            docs = np.ctypeslib.as_array(doc_off, (n + 1,)).copy()
            spans = np.ctypeslib.as_array(span, (2 * m,)).copy() if m else np.zeros(0, np.int64)
            offs = np.ctypeslib.as_array(text_off, (m + 1,)).copy()
            data = C.string_at(blob, int(offs[m])) if m else b""
        finally:
            self._lib.vr_chunks_free(handle)
        whole = data.decode("utf-8", "surrogatepass")
        offs, spans = offs.tolist(), spans.tolist()
        if len(whole) == len(data):  # pure ASCII: byte offsets are character offsets, slice the one decoded string
            texts_out = [whole[offs[i]:offs[i + 1]] for i in range(m)]
        else:
            texts_out = [data[offs[i]:offs[i + 1]].decode("utf-8", "surrogatepass") for i in range(m)]
        out: list[list[Chunk]] = []
        for d in range(n):
            lo, hi = int(docs[d]), int(docs[d + 1])
            out.append([Chunk(texts_out[i], i - lo, spans[2 * i], spans[2 * i + 1]) for i in range(lo, hi)])
        return out


def get_chunking_service() -> ChunkingService:
    """chunking.py:244-246."""
    return ChunkingService()
