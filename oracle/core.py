"""TEST INFRASTRUCTURE — NumPy-facing wrapper over oracle_core.c (see that file's header)."""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build())
        fp, i64p, i32p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                               C.POINTER(C.c_uint8))
        L.oracle_cosine_preprocess.argtypes = [fp, C.c_int64, C.c_int32, fp]
        L.oracle_cosine_preprocess.restype = None
        L.oracle_dense_scores.argtypes = [fp, C.c_int32, fp, C.c_int64, C.c_int32, fp]
        L.oracle_dense_scores.restype = None
        L.oracle_idf.argtypes = [C.c_int64, C.c_int32]
        L.oracle_idf.restype = C.c_float
        L.oracle_sparse_scores.argtypes = [i64p, i32p, fp, C.c_int64, i32p, fp, C.c_int32, i32p,
                                           C.c_int64, fp]
        L.oracle_sparse_scores.restype = None
        L.oracle_topk.argtypes = [fp, u8p, C.c_int64, C.c_int32, i64p, fp]
        L.oracle_topk.restype = C.c_int32
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def cosine_preprocess(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    n, d = x.shape
    lib().oracle_cosine_preprocess(_p(x, C.c_float), n, d, _p(out, C.c_float))
    return out


def dense_scores(q_hat: np.ndarray, x_hat: np.ndarray) -> np.ndarray:
    q_hat = np.ascontiguousarray(q_hat, dtype=np.float32)
    x_hat = np.ascontiguousarray(x_hat, dtype=np.float32)
    nq, d = q_hat.shape
    n = x_hat.shape[0]
    out = np.empty((nq, n), dtype=np.float32)
    lib().oracle_dense_scores(_p(q_hat, C.c_float), nq, _p(x_hat, C.c_float), n, d, _p(out, C.c_float))
    return out


def idf(n_points: int, df: int) -> float:
    return float(lib().oracle_idf(n_points, df))


def topk(scores: np.ndarray, k: int, mask: np.ndarray | None = None):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows = np.empty(k, dtype=np.int64)
    out = np.empty(k, dtype=np.float32)
    mp = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        mp = _p(mask, C.c_uint8)
    c = lib().oracle_topk(_p(scores, C.c_float), mp, scores.shape[0], k, _p(rows, C.c_int64),
                          _p(out, C.c_float))
    return rows[:c].copy(), out[:c].copy()


def to_csr(sparse_rows):
    """[(indices, values)] -> (off int64, idx int32, val f32) with every row sorted by index."""
    off = np.zeros(len(sparse_rows) + 1, dtype=np.int64)
    idx_l, val_l = [], []
    for i, (ix, vs) in enumerate(sparse_rows):
        ix = np.asarray(ix, dtype=np.int32)
        vs = np.asarray(vs, dtype=np.float32)
        order = np.argsort(ix, kind="stable")
        idx_l.append(ix[order])
        val_l.append(vs[order])
        off[i + 1] = off[i] + len(ix)
    idx = np.concatenate(idx_l) if idx_l else np.zeros(0, np.int32)
    val = np.concatenate(val_l) if val_l else np.zeros(0, np.float32)
    return off, idx.astype(np.int32), val.astype(np.float32)


def document_frequencies(sparse_rows, live=None):
    """Qdrant Modifier.IDF statistics: df per token id and N = points carrying a sparse vector."""
    df: dict[int, int] = {}
    n = 0
    for i, row in enumerate(sparse_rows):
        if row is None or (live is not None and not live[i]):
            continue
        n += 1
        for t in set(int(v) for v in row[0]):
            df[t] = df.get(t, 0) + 1
    return df, n


def sparse_scores(sparse_rows, q_idx, q_val, live=None) -> np.ndarray:
    """scores over all rows (-inf = shares no term); rows that are None carry no sparse vector."""
    rows = [r if r is not None else ([], []) for r in sparse_rows]
    off, idx, val = to_csr(rows)
    df, n_points = document_frequencies(sparse_rows, live)
    q = sorted({int(i): float(v) for i, v in reversed(list(zip(q_idx, q_val)))}.items())
    qi = np.array([i for i, _ in q], dtype=np.int32)
    qv = np.array([v for _, v in q], dtype=np.float32)
    qdf = np.array([df.get(int(i), 0) for i in qi], dtype=np.int32)
    out = np.empty(len(rows), dtype=np.float32)
    lib().oracle_sparse_scores(_p(off, C.c_int64), _p(idx, C.c_int32), _p(val, C.c_float), len(rows),
                               _p(qi, C.c_int32), _p(qv, C.c_float), len(qi), _p(qdf, C.c_int32),
                               n_points, _p(out, C.c_float))
    return out


class SparseOracle:
    """sparse_scores() for many queries over one collection state: the CSR form and the document frequencies
    (Qdrant Modifier.IDF statistic over the LIVE points) are derived once."""

    def __init__(self, sparse_rows, live=None):
        self.rows = [r if r is not None else ([], []) for r in sparse_rows]
        self.off, self.idx, self.val = to_csr(self.rows)
        self.df, self.n_points = document_frequencies(sparse_rows, live)

    def scores(self, q_idx, q_val) -> np.ndarray:
        q = sorted({int(i): float(v) for i, v in reversed(list(zip(q_idx, q_val)))}.items())
        qi = np.array([i for i, _ in q], dtype=np.int32)
        qv = np.array([v for _, v in q], dtype=np.float32)
        qdf = np.array([self.df.get(int(i), 0) for i in qi], dtype=np.int32)
        out = np.empty(len(self.rows), dtype=np.float32)
        lib().oracle_sparse_scores(_p(self.off, C.c_int64), _p(self.idx, C.c_int32), _p(self.val, C.c_float), len(self.rows),
                                   _p(qi, C.c_int32), _p(qv, C.c_float), len(qi), _p(qdf, C.c_int32),
                                   self.n_points, _p(out, C.c_float))
        return out

    @classmethod
    def from_csr(cls, off, idx, val, live=None):
        """The same from CSR arrays whose rows are already sorted by id (full-size tests: no Python loop per row)."""
        self = cls.__new__(cls)
        self.off = np.ascontiguousarray(off, np.int64)
        self.idx = np.ascontiguousarray(idx, np.int32)
        self.val = np.ascontiguousarray(val, np.float32)
        n = len(self.off) - 1
        self.rows = range(n)
        row_of = np.repeat(np.arange(n), np.diff(self.off))
        keep = np.ones(len(self.idx), bool) if live is None else np.asarray(live, bool)[row_of]
        pairs = np.unique(row_of[keep].astype(np.int64) << 31 | self.idx[keep].astype(np.int64))  # a term counts once per row
        ids, counts = np.unique(pairs & ((1 << 31) - 1), return_counts=True)
        self.df = dict(zip(ids.tolist(), counts.tolist()))
        has = np.diff(self.off) >= 0  # every row of a CSR batch carries a sparse vector (possibly empty)
        self.n_points = int(has.sum() if live is None else (has & np.asarray(live, bool)).sum())
        return self
