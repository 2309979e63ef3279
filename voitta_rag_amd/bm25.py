"""Host-side helpers of the BM25 sparse model: the text pipeline lives in C++ (bm25_text.cpp,
reached through vr_bm25_tokenize); no CPU arithmetic happens in Python."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def hashed_stems(texts: list[str]) -> tuple[np.ndarray, np.ndarray]:
    """texts -> (off int64[n+1], ids int32[total]): abs(murmur3) of every stemmed token, text order."""
    lib = _lib.load_library()
    n = len(texts)
    raw = [t.encode("utf-8", "surrogatepass") if isinstance(t, str) else bytes(t) for t in texts]
    arr = (C.c_char_p * max(n, 1))(*raw) if n else (C.c_char_p * 1)()
    lens = np.array([len(b) for b in raw], np.int64)
    cap = int(lens.sum() // 2 + n + 1)  # a token needs >= 1 byte plus a separator
    off = np.zeros(n + 1, np.int64)
    ids = np.zeros(max(cap, 1), np.int32)
    need = C.c_int64()
    for _ in range(2):
        rc = lib.vr_bm25_tokenize(arr, lens.ctypes.data_as(C.POINTER(C.c_int64)), n,
                                  off.ctypes.data_as(C.POINTER(C.c_int64)), ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                  ids.shape[0], C.byref(need))
        if rc != -2:
            break
        ids = np.zeros(need.value, np.int32)  # (cannot happen with the bound above; the ABI allows it)
    check(rc)
    return off, ids[: need.value].copy()


def stem(word: str) -> str:
    lib = _lib.load_library()
    b = word.encode("utf-8")
    out = C.create_string_buffer(len(b) + 8)
    check(lib.vr_porter2_stem(b, len(b), out, len(out)))
    return out.value.decode("utf-8")
