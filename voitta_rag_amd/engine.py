"""Thin, typed Python face of the C-ABI (include/voitta_engine.h). NumPy arrays are passed as
host pointers; anything exposing ``data_ptr()`` on the engine's device (a torch-ROCm tensor) is
passed as a device pointer. No arithmetic happens here."""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import VR_FUSION_MINMAX, VR_FUSION_RRF, VR_MEM_DEVICE, VR_MEM_HOST, VR_TS_ABSENT, check

__all__ = ["Engine", "SearchFilter", "VR_FUSION_MINMAX", "VR_FUSION_RRF", "VR_TS_ABSENT"]


def _is_device_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and bool(x.is_cuda)


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class SearchFilter:
    """Integer form of VectorStoreService._build_filter's arguments
    (reference: src/voitta/services/vector_store.py:462-530); ids are dictionary ids of
    folder_path / index_folder strings. ``None`` for a must-set means "no such condition"."""

    folder_filter: int | None = None
    include_folders: list[int] | None = None
    exclude_folders: list[int] = field(default_factory=list)
    exclude_index_folders: list[int] = field(default_factory=list)
    date_start: int | None = None
    date_end: int | None = None
    date_field: str | None = None  # "created" | "modified" | None

    def is_empty(self) -> bool:
        return (self.folder_filter is None and self.include_folders is None and not self.exclude_folders
                and not self.exclude_index_folders and self.date_start is None and self.date_end is None)

    def to_c(self):
        """-> (VrFilter, keepalive) ; keepalive holds the arrays the struct points into."""
        sets = []
        if self.folder_filter is not None:
            sets.append([int(self.folder_filter)])
        if self.include_folders is not None:
            sets.append([int(v) for v in self.include_folders])
        ids = _np([v for s in sets for v in s], np.int32)
        off = _np(np.cumsum([0] + [len(s) for s in sets]), np.int32)
        nf = _np(self.exclude_folders, np.int32)
        nif = _np(self.exclude_index_folders, np.int32)
        f = _lib.VrFilter()
        f.struct_size = C.sizeof(_lib.VrFilter)
        f.n_must_folder_sets = len(sets)
        f.must_folder_ids = _ptr(ids, C.c_int32)
        f.must_folder_off = _ptr(off, C.c_int32)
        f.not_folder_ids = _ptr(nf, C.c_int32)
        f.n_not_folder = len(nf)
        f.n_not_index_folder = len(nif)
        f.not_index_folder_ids = _ptr(nif, C.c_int32)
        f.has_date_start = int(self.date_start is not None)
        f.has_date_end = int(self.date_end is not None)
        f.date_start = int(self.date_start or 0)
        f.date_end = int(self.date_end or 0)
        # field_map.get(date_field, "source_modified_at"), vector_store.py:511-512
        f.date_field = 1 if self.date_field == "created" else 0
        return f, (ids, off, nf, nif)


class Engine:
    """One HBM-resident index on one GPU."""

    def __init__(self, dim: int, device: int = 0, initial_rows: int = 0, prefilter: bool = True):
        """prefilter: keep the int8 (f16 when dim % 64 != 0 or VR_PREFILTER=f16) shadow corpus and run single-query dense searches in two
        stages (f16 scan + exact re-score; identical results, half of the bytes)."""
        self._lib = _lib.load_library()
        cfg = _lib.VrConfig()
        cfg.struct_size = C.sizeof(_lib.VrConfig)
        cfg.device = device
        cfg.dim = dim
        cfg.flags = 0 if prefilter else 1  # VR_ENGINE_NO_PREFILTER
        cfg.initial_rows = initial_rows
        h = C.c_void_p()
        check(self._lib.vr_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.dim = dim
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vr_engine_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing -------------------------------------------------------------------------
    @property
    def handle(self):
        return self._h

    def sync(self) -> None:
        check(self._lib.vr_sync(self._h))

    def stream_ptr(self) -> int:
        return int(self._lib.vr_stream(self._h) or 0)

    def set_stream(self, stream_ptr: int | None) -> None:
        """Queue all further engine work on this hipStream_t (0/None = the engine's own stream)."""
        check(self._lib.vr_set_stream(self._h, C.c_void_p(stream_ptr or None)))
        self._bound = int(stream_ptr or 0)

    def _follow(self, tensor) -> None:
        """Device tensors are produced on the framework's current stream: run there too."""
        import torch

        ptr = int(torch.cuda.current_stream(tensor.device).cuda_stream)
        if ptr != getattr(self, "_bound", 0):
            self.set_stream(ptr)

    # ---- measurement ----------------------------------------------------------------------
    PROF_GEMM, PROF_ATTENTION, PROF_DENSE_SCAN, PROF_SPARSE_SCAN, PROF_BATCH_SCAN = 0, 1, 2, 3, 4

    def profile(self, enable: bool) -> None:
        """HIP-event timing of the engine's kernels on its own stream (vr_profile)."""
        check(self._lib.vr_profile(self._h, int(enable)))

    def profile_read(self, kernel_class: int) -> tuple[float, int, float]:
        """-> (total milliseconds, launches, total algorithmic work: FLOP or bytes)"""
        ms, n, w = C.c_double(), C.c_int64(), C.c_double()
        check(self._lib.vr_profile_read(self._h, kernel_class, C.byref(ms), C.byref(n), C.byref(w)))
        return float(ms.value), int(n.value), float(w.value)

    def idf(self, n_points: int, df: int) -> float:
        return float(self._lib.vr_idf(int(n_points), int(df)))

    # ---- store ----------------------------------------------------------------------------
    def upsert(self, dense, sparse=None, folder_ids=None, index_folder_ids=None, created=None,
               modified=None) -> int:
        """dense: (n, D) f32 NumPy array or device tensor. sparse: None, or a list of
        (indices, values) per row (host), or a CSR triple (off, idx, val) of NumPy arrays /
        device tensors matching ``dense``'s memory space. Returns the first assigned row."""
        if _is_device_tensor(dense):
            self._follow(dense)
            mem = VR_MEM_DEVICE
            n = int(dense.shape[0])
            assert dense.is_contiguous() and tuple(dense.shape)[1] == self.dim
            dptr = C.c_void_p(dense.data_ptr())
            keep = [dense]
        else:
            mem = VR_MEM_HOST
            d = _np(dense, np.float32).reshape(-1, self.dim)
            n = d.shape[0]
            dptr = C.c_void_p(d.ctypes.data)
            keep = [d]
        op = ip = vp = C.c_void_p(None)
        if sparse is not None:
            if isinstance(sparse, tuple) and len(sparse) == 3 and not isinstance(sparse[0], (list, tuple)):
                off, idx, val = sparse
            else:
                off = np.zeros(n + 1, np.int64)
                for i, (ix, _) in enumerate(sparse):
                    off[i + 1] = off[i] + len(ix)
                idx = np.concatenate([_np(ix, np.int32) for ix, _ in sparse]) if n else np.zeros(0, np.int32)
                val = np.concatenate([_np(vs, np.float32) for _, vs in sparse]) if n else np.zeros(0, np.float32)
            if mem == VR_MEM_DEVICE:
                op, ip, vp = (C.c_void_p(t.data_ptr()) for t in (off, idx, val))
                keep += [off, idx, val]
            else:
                off, idx, val = _np(off, np.int64), _np(idx, np.int32), _np(val, np.float32)
                assert off.shape[0] == n + 1
                op, ip, vp = (C.c_void_p(a.ctypes.data) for a in (off, idx, val))
                keep += [off, idx, val]

        def col(a, dtype, ctype):
            if a is None:
                return None
            a = _np(a, dtype)
            assert a.shape == (n,)
            keep.append(a)
            return _ptr(a, ctype)

        first = C.c_int64(-1)
        check(self._lib.vr_upsert(self._h, n, mem, dptr, op, ip, vp,
                                  col(folder_ids, np.int32, C.c_int32),
                                  col(index_folder_ids, np.int32, C.c_int32),
                                  col(created, np.int64, C.c_int64), col(modified, np.int64, C.c_int64),
                                  C.byref(first)))
        return int(first.value)

    # ---- BM25 / fused indexing ------------------------------------------------------------
    BM25_K, BM25_B, BM25_AVG_LEN = 1.2, 0.75, 256.0  # fastembed Qdrant/bm25 defaults [EXT]

    def bm25_tf(self, tok_off, tok_ids, k: float = BM25_K, b: float = BM25_B, avg_len: float = BM25_AVG_LEN):
        """Hashed stems per document (host arrays) -> [(indices int32 asc, values f64)] per document."""
        off = _np(tok_off, np.int64)
        ids = _np(tok_ids, np.int32)
        n = off.shape[0] - 1
        cnt = np.zeros(max(n, 0), np.int32)
        idx = np.zeros(max(int(off[-1]) if n >= 0 else 0, 1), np.int32)
        val = np.zeros(idx.shape[0], np.float64)
        check(self._lib.vr_bm25_tf(self._h, C.c_void_p(off.ctypes.data), C.c_void_p(ids.ctypes.data), n, VR_MEM_HOST,
                                   k, b, avg_len, C.c_void_p(cnt.ctypes.data), C.c_void_p(idx.ctypes.data),
                                   C.c_void_p(val.ctypes.data)))
        return [(idx[off[d]: off[d] + cnt[d]].copy(), val[off[d]: off[d] + cnt[d]].copy()) for d in range(n)]

    def index_batch(self, wp_ids, wp_off, bm_ids=None, bm_off=None, folder_ids=None, index_folder_ids=None,
                    created=None, modified=None, k: float = BM25_K, b: float = BM25_B,
                    avg_len: float = BM25_AVG_LEN) -> int:
        """encode -> BM25 tf -> store, all on the GPU. Token arrays are NumPy (host) or device tensors."""
        dev = _is_device_tensor(wp_ids)
        if dev:
            self._follow(wp_ids)
            n = int(wp_off.shape[0]) - 1
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)  # noqa: E731
            keep = [wp_ids, wp_off, bm_ids, bm_off]
            args = [ptr(wp_ids), ptr(wp_off), ptr(bm_ids), ptr(bm_off)]
        else:
            wp_ids, wp_off = _np(wp_ids, np.int32), _np(wp_off, np.int32)
            n = wp_off.shape[0] - 1
            if bm_ids is not None:
                bm_ids, bm_off = _np(bm_ids, np.int32), _np(bm_off, np.int64)
            keep = [wp_ids, wp_off, bm_ids, bm_off]
            args = [C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(None) for a in keep]

        def col(a, dtype, ctype):
            if a is None:
                return None
            a = _np(a, dtype)
            assert a.shape == (n,)
            keep.append(a)
            return _ptr(a, ctype)

        first = C.c_int64(-1)
        check(self._lib.vr_index_batch(self._h, n, VR_MEM_DEVICE if dev else VR_MEM_HOST, *args, k, b, avg_len,
                                       col(folder_ids, np.int32, C.c_int32), col(index_folder_ids, np.int32, C.c_int32),
                                       col(created, np.int64, C.c_int64), col(modified, np.int64, C.c_int64),
                                       C.byref(first)))
        return int(first.value)

    def delete_rows(self, rows) -> None:
        r = _np(rows, np.int64)
        check(self._lib.vr_delete_rows(self._h, _ptr(r, C.c_int64), r.shape[0]))

    def stats(self) -> dict:
        """Counters of the two-stage dense search: searches served, overflow fallbacks, last candidate count."""
        out = {}
        for name, which in (("two_stage", 0), ("fallback", 1), ("last_candidates", 2), ("batched", 3),
                            ("batch_fallback", 4), ("batch_candidates", 6), ("sparse_grouped", 7),
                            ("sparse_group_redo", 8), ("sparse_group_candidates", 9)):
            v = C.c_int64()
            check(self._lib.vr_stats(self._h, which, C.byref(v)))
            out[name] = int(v.value)
        return out

    def generation(self) -> int:
        """Bumped whenever row numbers change meaning (compaction, load): host tables keyed by row belong to one."""
        v = C.c_int64()
        check(self._lib.vr_stats(self._h, 5, C.byref(v)))
        return int(v.value)

    def compact(self) -> np.ndarray:
        """Drop tombstoned rows; surviving rows are renumbered in order. Returns new_row_of_old
        (int64, -1 for dropped rows) so the caller can renumber whatever it keys by row."""
        n_rows, _ = self.count()
        remap = np.empty(n_rows, np.int64)
        after = C.c_int64()
        check(self._lib.vr_compact(self._h, _ptr(remap, C.c_int64), C.byref(after)))
        return remap

    def save(self, path: str) -> None:
        """Write the device-resident index (dense, sparse, payload columns, tombstones, document
        frequencies) to one checksummed file; the write is atomic (tmp + rename)."""
        check(self._lib.vr_save(self._h, os.fsencode(path)))

    def load(self, path: str) -> None:
        """Restore a file written by save() into this (empty, same-dimension) engine."""
        check(self._lib.vr_load(self._h, os.fsencode(path)))

    def count(self) -> tuple[int, int]:
        a, b = C.c_int64(), C.c_int64()
        check(self._lib.vr_count(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def get_dense(self, rows) -> np.ndarray:
        r = _np(rows, np.int64)
        out = np.empty((r.shape[0], self.dim), np.float32)
        check(self._lib.vr_get_dense(self._h, _ptr(r, C.c_int64), r.shape[0], _ptr(out, C.c_float)))
        return out

    def sparse_stats(self, ids) -> tuple[np.ndarray, int]:
        i = _np(ids, np.int32)
        df = np.zeros(i.shape[0], np.int32)
        n = C.c_int64()
        check(self._lib.vr_sparse_stats(self._h, _ptr(i, C.c_int32), i.shape[0], _ptr(df, C.c_int32), C.byref(n)))
        return df, int(n.value)

    # ---- search ---------------------------------------------------------------------------
    @staticmethod
    def _filter(flt: SearchFilter | None):
        if flt is None or flt.is_empty():
            return None, None
        f, keep = flt.to_c()
        return C.byref(f), (f, keep)

    def search_dense(self, queries, k: int, flt: SearchFilter | None = None, raw: bool = False):
        """-> list of (rows int64[c], scores f32[c]) per query; raw=True: the (nq, k) row / score arrays (-1 / 0 padded)
        and the counts as the C-ABI fills them (a 1000-query batch spends a tenth of its time being cut into 2000 small
        arrays otherwise)."""
        if _is_device_tensor(queries):
            self._follow(queries)
            mem, nq = VR_MEM_DEVICE, int(queries.shape[0])
            qp = C.c_void_p(queries.data_ptr())
        else:
            q = _np(queries, np.float32).reshape(-1, self.dim)
            mem, nq = VR_MEM_HOST, q.shape[0]
            qp = C.c_void_p(q.ctypes.data)
        rows = np.empty((nq, k), np.int64)
        scores = np.empty((nq, k), np.float32)
        counts = np.zeros(nq, np.int32)
        fp, keep = self._filter(flt)
        check(self._lib.vr_search_dense(self._h, qp, nq, mem, k, fp, _ptr(rows, C.c_int64),
                                        _ptr(scores, C.c_float), _ptr(counts, C.c_int32)))
        del keep
        if raw:
            return rows, scores, counts
        return [(rows[i, : counts[i]].copy(), scores[i, : counts[i]].copy()) for i in range(nq)]

    def search_dense_keys(self, queries, k: int, flt: SearchFilter | None = None, out=None):
        """The dense search with results as packed ranking keys (include/voitta_engine.h): an (nq, k) uint64 NumPy
        array, or — when ``out`` is an (nq, k) int64 device tensor — written there without leaving the device."""
        if _is_device_tensor(queries):
            self._follow(queries)
            mem, nq = VR_MEM_DEVICE, int(queries.shape[0])
            qp = C.c_void_p(queries.data_ptr())
        else:
            q = _np(queries, np.float32).reshape(-1, self.dim)
            mem, nq = VR_MEM_HOST, q.shape[0]
            qp = C.c_void_p(q.ctypes.data)
        fp, keep = self._filter(flt)
        if out is not None:
            assert _is_device_tensor(out) and out.is_contiguous() and tuple(out.shape) == (nq, k) and out.element_size() == 8
            self._follow(out)
            check(self._lib.vr_search_dense_keys(self._h, qp, nq, mem, k, fp, C.c_void_p(out.data_ptr()), VR_MEM_DEVICE))
            return out
        keys = np.empty((nq, k), np.uint64)
        check(self._lib.vr_search_dense_keys(self._h, qp, nq, mem, k, fp, C.c_void_p(keys.ctypes.data), VR_MEM_HOST))
        del keep
        return keys

    @staticmethod
    def decode_keys(keys: np.ndarray):
        """(nq, k) uint64 keys -> (rows int64 with -1 for empty slots, scores f32)."""
        keys = np.asarray(keys, np.uint64)
        hi = (keys >> np.uint64(32)).astype(np.uint32)
        bits = np.where(hi & np.uint32(0x80000000), hi ^ np.uint32(0x80000000), ~hi)
        rows = np.int64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        rows[keys == 0] = -1
        scores = bits.astype(np.uint32).view(np.float32).copy()
        scores[keys == 0] = 0.0
        return rows, scores

    def search_sparse(self, q_idx, q_val, k: int, flt: SearchFilter | None = None, weights_given: bool = False):
        """weights_given: q_val already holds q_t * idf_t (sharded search with global statistics)."""
        qi, qv = _np(q_idx, np.int32), _np(q_val, np.float32)
        rows = np.empty(k, np.int64)
        scores = np.empty(k, np.float32)
        c = C.c_int32()
        fp, keep = self._filter(flt)
        check(self._lib.vr_search_sparse(self._h, _ptr(qi, C.c_int32), _ptr(qv, C.c_float), qi.shape[0], k,
                                         int(weights_given), fp, _ptr(rows, C.c_int64), _ptr(scores, C.c_float),
                                         C.byref(c)))
        del keep
        return rows[: c.value].copy(), scores[: c.value].copy()

    def search_hybrid(self, query, q_idx, q_val, limit: int, sparse_weight: float = 0.1,
                      fusion: int = VR_FUSION_MINMAX, flt: SearchFilter | None = None):
        """-> (rows int64[c], fused scores f64[c], from_dense int32[c])"""
        if _is_device_tensor(query):
            self._follow(query)
            mem, qp = VR_MEM_DEVICE, C.c_void_p(query.data_ptr())
        else:
            q = _np(query, np.float32).reshape(self.dim)
            mem, qp = VR_MEM_HOST, C.c_void_p(q.ctypes.data)
        qi, qv = _np(q_idx, np.int32), _np(q_val, np.float32)
        # latency path: output buffers and their addresses are kept per `limit`, and the call goes through a
        # prototype that takes plain addresses (building five ctypes POINTER objects per query cost ~10 us).
        # The buffers are PER THREAD: the engine call drops the GIL and an engine may be called from any number
        # of threads (include/voitta_engine.h), so a shared set could be overwritten before it is copied out.
        if not hasattr(self, "_hybrid_fn"):
            fn = self._lib["vr_search_hybrid"]
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double,
                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            self._hybrid_fn = fn
            self._hybrid_tls = threading.local()
        slots = getattr(self._hybrid_tls, "slots", None)
        if slots is None:
            slots = self._hybrid_tls.slots = {}
        slot = slots.get(limit)
        if slot is None:
            rows, scores, fd = np.empty(limit, np.int64), np.empty(limit, np.float64), np.empty(limit, np.int32)
            cnt = np.zeros(1, np.int32)
            slot = (rows, scores, fd, cnt, rows.ctypes.data, scores.ctypes.data, fd.ctypes.data, cnt.ctypes.data)
            slots[limit] = slot
        rows, scores, fd, cnt, rp, sp, fdp, cp = slot
        if flt is None or flt.is_empty():
            fptr, keep = None, None
        else:
            f, arrays = flt.to_c()
            fptr, keep = C.addressof(f), (f, arrays)
        check(self._hybrid_fn(self._h, qp, mem, qi.ctypes.data, qv.ctypes.data, qi.shape[0], limit,
                              float(sparse_weight), fusion, fptr, rp, sp, fdp, cp))
        del keep
        n = int(cnt[0])
        return rows[:n].copy(), scores[:n].copy(), fd[:n].copy()

    def query_text(self, tokenizer_handle, dense_text: str, sparse_text: str | None, max_len: int, limit: int,
                   sparse_weight: float = 0.1, fusion: int = VR_FUSION_MINMAX, flt: SearchFilter | None = None):
        """A question as text in ONE engine call (vr_query_text): WordPiece + BM25 tokenise + encode + hybrid (or, when
        no stem survives, dense) search. -> (rows int64[c], scores f64[c], from_dense int32[c], hybrid: bool)."""
        d = dense_text.encode("utf-8", "replace")
        sp = sparse_text.encode("utf-8", "surrogatepass") if sparse_text else None
        rows, scores, fd = np.empty(limit, np.int64), np.empty(limit, np.float64), np.empty(limit, np.int32)
        cnt, hyb = C.c_int32(), C.c_int32()
        fp, keep = self._filter(flt)
        check(self._lib.vr_query_text(self._h, tokenizer_handle, d, len(d), sp, len(sp) if sp else 0, int(max_len), limit,
                                      float(sparse_weight), fusion, fp, _ptr(rows, C.c_int64), _ptr(scores, C.c_double),
                                      _ptr(fd, C.c_int32), C.byref(cnt), C.byref(hyb)))
        del keep
        n = int(cnt.value)
        return rows[:n], scores[:n], fd[:n], bool(hyb.value)

    # ---- many queries per call (BASELINE configs[4]: 1k batched hybrid queries) -----------------------------------
    @staticmethod
    def _sparse_csr(sparse_queries, nq: int):
        """[(indices, values)] per query (None / empty = no sparse terms) -> CSR (off int64, idx int32, val f32).
        A caller that already holds the batch as CSR passes the triple (off, idx, val) itself."""
        if isinstance(sparse_queries, tuple) and len(sparse_queries) == 3 and not isinstance(sparse_queries[0], (list, tuple)):
            off, idx, val = (_np(sparse_queries[0], np.int64), _np(sparse_queries[1], np.int32), _np(sparse_queries[2], np.float32))
            assert off.shape == (nq + 1,) and idx.shape == val.shape == (int(off[-1]),)
            return off, idx, val
        assert len(sparse_queries) == nq
        off = np.zeros(nq + 1, np.int64)
        idx, val = [], []
        for i, sq in enumerate(sparse_queries):
            n = 0
            if sq is not None and len(sq[0]) > 0:
                qi, qv = _np(sq[0], np.int32).reshape(-1), _np(sq[1], np.float32).reshape(-1)
                assert qi.shape == qv.shape
                idx.append(qi)
                val.append(qv)
                n = qi.shape[0]
            off[i + 1] = off[i] + n
        idx = np.concatenate(idx) if idx else np.zeros(0, np.int32)
        val = np.concatenate(val) if val else np.zeros(0, np.float32)
        return off, _np(idx, np.int32), _np(val, np.float32)

    def _queries(self, queries):
        if _is_device_tensor(queries):
            self._follow(queries)
            assert queries.is_contiguous() and int(queries.shape[-1]) == self.dim
            return VR_MEM_DEVICE, int(queries.shape[0]), C.c_void_p(queries.data_ptr()), queries
        q = _np(queries, np.float32).reshape(-1, self.dim)
        return VR_MEM_HOST, q.shape[0], C.c_void_p(q.ctypes.data), q

    def search_sparse_batch(self, sparse_queries, k: int, flt: SearchFilter | None = None, weights_given: bool = False):
        """nq sparse searches in one call -> list of (rows int64[c], scores f32[c]); bit for bit nq x search_sparse."""
        csr = isinstance(sparse_queries, tuple) and len(sparse_queries) == 3 and not isinstance(sparse_queries[0], (list, tuple))
        nq = len(sparse_queries[0]) - 1 if csr else len(sparse_queries)
        if nq == 0:
            return []
        off, idx, val = self._sparse_csr(sparse_queries, nq)
        rows = np.empty((nq, k), np.int64)
        scores = np.empty((nq, k), np.float32)
        counts = np.zeros(nq, np.int32)
        fp, keep = self._filter(flt)
        check(self._lib.vr_search_sparse_batch(self._h, _ptr(off, C.c_int64), _ptr(idx, C.c_int32), _ptr(val, C.c_float), nq, k,
                                               int(weights_given), fp, _ptr(rows, C.c_int64), _ptr(scores, C.c_float),
                                               _ptr(counts, C.c_int32)))
        del keep
        return [(rows[i, : counts[i]].copy(), scores[i, : counts[i]].copy()) for i in range(nq)]

    def search_hybrid_batch(self, queries, sparse_queries, limit: int, sparse_weight: float = 0.1,
                            fusion: int = VR_FUSION_MINMAX, flt: SearchFilter | None = None, raw: bool = False):
        """nq hybrid searches in ONE call (one batched dense search, one batched sparse search beside it, fusion on
        the host threads) -> list of (rows int64[c], fused scores f64[c], from_dense int32[c]) per query, bit for bit
        what nq x search_hybrid return. raw=True: the (nq, limit) arrays and the counts instead of the list."""
        mem, nq, qp, keepq = self._queries(queries)
        off, idx, val = self._sparse_csr(sparse_queries, nq)
        rows = np.empty((nq, limit), np.int64)
        scores = np.empty((nq, limit), np.float64)
        fd = np.empty((nq, limit), np.int32)
        counts = np.zeros(nq, np.int32)
        fp, keep = self._filter(flt)
        check(self._lib.vr_search_hybrid_batch(self._h, qp, nq, mem, _ptr(off, C.c_int64), _ptr(idx, C.c_int32),
                                               _ptr(val, C.c_float), limit, float(sparse_weight), fusion, fp,
                                               _ptr(rows, C.c_int64), _ptr(scores, C.c_double), _ptr(fd, C.c_int32),
                                               _ptr(counts, C.c_int32)))
        del keep, keepq
        if raw:
            return rows, scores, fd, counts
        return [(rows[i, : counts[i]].copy(), scores[i, : counts[i]].copy(), fd[i, : counts[i]].copy()) for i in range(nq)]

    def search_hybrid_keys(self, queries, sparse_queries, k: int, flt: SearchFilter | None = None,
                           weights_given: bool = False, out=None):
        """Both legs of nq hybrid queries as packed ranking keys, (nq, 2, k) uint64: per query its dense list, then
        its sparse list. ``out``: an (nq, 2, k) int64 device tensor to write them into (the tensor a sharded caller
        hands to its all_gather), else a NumPy array is returned."""
        mem, nq, qp, keepq = self._queries(queries)
        off, idx, val = self._sparse_csr(sparse_queries, nq)
        fp, keep = self._filter(flt)
        args = (self._h, qp, nq, mem, _ptr(off, C.c_int64), _ptr(idx, C.c_int32), _ptr(val, C.c_float), k, int(weights_given), fp)
        if out is not None:
            assert _is_device_tensor(out) and out.is_contiguous() and tuple(out.shape) == (nq, 2, k) and out.element_size() == 8
            self._follow(out)
            check(self._lib.vr_search_hybrid_keys(*args, C.c_void_p(out.data_ptr()), VR_MEM_DEVICE))
            return out
        keys = np.empty((nq, 2, k), np.uint64)
        check(self._lib.vr_search_hybrid_keys(*args, C.c_void_p(keys.ctypes.data), VR_MEM_HOST))
        del keep, keepq
        return keys

    def merge_keys(self, parts, k: int):
        """parts: (n_parts, n_lists, k) packed keys — a NumPy uint64 array or an int64 device tensor (what an all_gather
        of search_dense_keys / search_hybrid_keys outputs filled). -> (global ids int64 (n_lists, k), -1 padded;
        scores f32 (n_lists, k); counts int32 (n_lists,)), merged per list: score descending, then ascending global id
        row * n_parts + part."""
        if _is_device_tensor(parts):
            self._follow(parts)
            assert parts.is_contiguous() and parts.element_size() == 8
            n_parts, n_lists = int(parts.shape[0]), int(np.prod(tuple(parts.shape)[1:-1]))
            mem, pp, keep = VR_MEM_DEVICE, C.c_void_p(parts.data_ptr()), parts
        else:
            a = np.ascontiguousarray(parts).view(np.uint64)
            n_parts, n_lists = a.shape[0], int(np.prod(a.shape[1:-1]))
            mem, pp, keep = VR_MEM_HOST, C.c_void_p(a.ctypes.data), a
        assert int(parts.shape[-1]) == k
        ids = np.empty((n_lists, k), np.int64)
        scores = np.empty((n_lists, k), np.float32)
        counts = np.zeros(n_lists, np.int32)
        check(self._lib.vr_merge_keys(self._h, pp, n_parts, n_lists, k, mem, _ptr(ids, C.c_int64), _ptr(scores, C.c_float),
                                      _ptr(counts, C.c_int32)))
        del keep
        return ids, scores, counts

    # ---- collection-wide document frequencies on a sharded corpus ----------------------------------------------------
    def sparse_row_ids(self, rows, device: bool = False):
        """Term ids of the listed rows' sparse vectors, (n, stride) int32 with -1 padding (dead rows and rows without
        a sparse vector: all -1), and the number of listed rows that are live and carry one. device=True: a torch
        tensor on the engine's device."""
        r = _np(rows, np.int64)
        stride, pts = C.c_int32(), C.c_int64()
        check(self._lib.vr_sparse_row_ids(self._h, _ptr(r, C.c_int64), r.shape[0], None, 0, VR_MEM_HOST, C.byref(stride), C.byref(pts)))
        w = int(stride.value)
        if device:
            import torch

            out = torch.full((r.shape[0], w), -1, dtype=torch.int32, device=torch.device("cuda", self.device))
            if out.numel():
                self._follow(out)
                check(self._lib.vr_sparse_row_ids(self._h, _ptr(r, C.c_int64), r.shape[0], C.c_void_p(out.data_ptr()), out.numel(),
                                                  VR_MEM_DEVICE, C.byref(stride), C.byref(pts)))
            return out, int(pts.value)
        out = np.full((r.shape[0], w), -1, np.int32)
        if out.size:
            check(self._lib.vr_sparse_row_ids(self._h, _ptr(r, C.c_int64), r.shape[0], C.c_void_p(out.ctypes.data), out.size,
                                              VR_MEM_HOST, C.byref(stride), C.byref(pts)))
        return out, int(pts.value)

    def df_apply(self, ids, n_points: int, sign: int = 1) -> None:
        """Add (sign = +1) or remove (-1) the statistics of rows stored on OTHER shards: df[id] += sign per id >= 0,
        sparse point count += sign * n_points."""
        if _is_device_tensor(ids):
            self._follow(ids)
            assert ids.is_contiguous() and ids.element_size() == 4
            check(self._lib.vr_df_apply(self._h, C.c_void_p(ids.data_ptr()), int(ids.numel()), VR_MEM_DEVICE, int(n_points), int(sign)))
            return
        a = _np(ids, np.int32).reshape(-1)
        check(self._lib.vr_df_apply(self._h, C.c_void_p(a.ctypes.data), a.shape[0], VR_MEM_HOST, int(n_points), int(sign)))


def fuse_minmax(d_rows, d_scores, s_rows, s_scores, limit: int, sparse_weight: float, json_scores: bool = True):
    """Host-only: the arithmetic of _hybrid_search on two result lists (no GPU needed)."""
    lib = _lib.load_library()
    dr, ds = _np(d_rows, np.int64), _np(d_scores, np.float32)
    sr, ss = _np(s_rows, np.int64), _np(s_scores, np.float32)
    cap = max(limit, 1)
    rows = np.empty(cap, np.int64)
    scores = np.empty(cap, np.float64)
    fd = np.empty(cap, np.int32)
    c = C.c_int32()
    check(lib.vr_fuse_minmax(_ptr(dr, C.c_int64), _ptr(ds, C.c_float), dr.shape[0], _ptr(sr, C.c_int64),
                             _ptr(ss, C.c_float), sr.shape[0], limit, float(sparse_weight), int(json_scores),
                             _ptr(rows, C.c_int64), _ptr(scores, C.c_double), _ptr(fd, C.c_int32), C.byref(c)))
    return rows[: c.value].copy(), scores[: c.value].copy(), fd[: c.value].copy()


def fuse_rrf(d_rows, s_rows, limit: int):
    """Host-only: reciprocal-rank fusion of two ranked row lists (vr_search_hybrid's VR_FUSION_RRF mode alone)."""
    lib = _lib.load_library()
    dr, sr = _np(d_rows, np.int64), _np(s_rows, np.int64)
    cap = max(limit, 1)
    rows = np.empty(cap, np.int64)
    scores = np.empty(cap, np.float64)
    fd = np.empty(cap, np.int32)
    c = C.c_int32()
    check(lib.vr_fuse_rrf(_ptr(dr, C.c_int64), dr.shape[0], _ptr(sr, C.c_int64), sr.shape[0], limit,
                          _ptr(rows, C.c_int64), _ptr(scores, C.c_double), _ptr(fd, C.c_int32), C.byref(c)))
    return rows[: c.value].copy(), scores[: c.value].copy(), fd[: c.value].copy()


def fuse_batch(d_ids, d_scores, d_counts, s_ids, s_scores, s_counts, limit: int, sparse_weight: float,
               fusion: int = VR_FUSION_MINMAX, json_scores: bool = True):
    """Host-only: fusion of nq pairs of lists ((nq, k) arrays + counts) on the host threads ->
    (rows (nq, limit) int64, scores (nq, limit) f64, from_dense (nq, limit) int32, counts (nq,) int32)."""
    lib = _lib.load_library()
    dr, ds, dc = _np(d_ids, np.int64), _np(d_scores, np.float32), _np(d_counts, np.int32)
    nq, k = dr.shape
    sr, ss, sc = _np(s_ids, np.int64), _np(s_scores, np.float32), _np(s_counts, np.int32)
    assert sr.shape == (nq, k) and ds.shape == (nq, k) and ss.shape == (nq, k)
    rows = np.empty((nq, limit), np.int64)
    scores = np.empty((nq, limit), np.float64)
    fd = np.empty((nq, limit), np.int32)
    counts = np.zeros(nq, np.int32)
    check(lib.vr_fuse_batch(_ptr(dr, C.c_int64), _ptr(ds, C.c_float), _ptr(dc, C.c_int32), _ptr(sr, C.c_int64),
                            _ptr(ss, C.c_float), _ptr(sc, C.c_int32), nq, k, limit, float(sparse_weight), fusion,
                            int(json_scores), _ptr(rows, C.c_int64), _ptr(scores, C.c_double), _ptr(fd, C.c_int32),
                            _ptr(counts, C.c_int32)))
    return rows, scores, fd, counts
