"""TEST INFRASTRUCTURE — a stand-in for voitta_rag_amd.Engine whose arithmetic is the CPU oracle (oracle/core.py,
oracle/fusion.py). It lets the host logic above the C-ABI (VectorStoreService, ShardedSearcher,
ShardedVectorStore) run in the no-GPU test tier and in multi-process gloo tests; nothing in the product imports it.
Only the methods those classes call exist."""
import numpy as np

from oracle import core as ocore
from oracle import fusion as ofus

TS_ABSENT = -(2 ** 63)


class OracleEngine:
    def __init__(self, dim, device=0, initial_rows=0, prefilter=True):
        self.dim = dim
        self.x = np.zeros((0, dim), np.float32)
        self.sp = []
        self.live = np.zeros(0, bool)
        self.folder = np.zeros(0, np.int32)
        self.ifolder = np.zeros(0, np.int32)
        self.created = np.zeros(0, np.int64)
        self.modified = np.zeros(0, np.int64)

    def close(self):
        pass

    def count(self):
        return len(self.live), int(self.live.sum())

    def upsert(self, dense, sparse=None, folder_ids=None, index_folder_ids=None, created=None, modified=None):
        d = ocore.cosine_preprocess(np.ascontiguousarray(dense, np.float32).reshape(-1, self.dim))
        n, first = d.shape[0], len(self.live)
        self.x = np.concatenate([self.x, d])
        self.sp += [None] * n if sparse is None else [
            (np.asarray(i, np.int32)[np.argsort(np.asarray(i, np.int32), kind="stable")],
             np.asarray(v, np.float32)[np.argsort(np.asarray(i, np.int32), kind="stable")]) for i, v in sparse]
        self.live = np.concatenate([self.live, np.ones(n, bool)])
        col = lambda a, dt, fill: np.full(n, fill, dt) if a is None else np.asarray(a, dt)  # noqa: E731
        self.folder = np.concatenate([self.folder, col(folder_ids, np.int32, 0)])
        self.ifolder = np.concatenate([self.ifolder, col(index_folder_ids, np.int32, 0)])
        self.created = np.concatenate([self.created, col(created, np.int64, TS_ABSENT)])
        self.modified = np.concatenate([self.modified, col(modified, np.int64, TS_ABSENT)])
        return first

    def delete_rows(self, rows):
        self.live[np.asarray(rows, np.int64)] = False

    def get_dense(self, rows):
        return self.x[np.asarray(rows, np.int64)]

    def _mask(self, flt):
        m = self.live.copy()
        if flt is None or flt.is_empty():
            return m
        if flt.folder_filter is not None:
            m &= self.folder == flt.folder_filter
        if flt.include_folders is not None:
            m &= np.isin(self.folder, flt.include_folders)
        if flt.exclude_folders:
            m &= ~np.isin(self.folder, flt.exclude_folders)
        if flt.exclude_index_folders:
            m &= ~np.isin(self.ifolder, flt.exclude_index_folders)
        if flt.date_start is not None or flt.date_end is not None:
            t = self.created if flt.date_field == "created" else self.modified
            m &= t != TS_ABSENT
            if flt.date_start is not None:
                m &= t >= flt.date_start
            if flt.date_end is not None:
                m &= t <= flt.date_end
        return m

    def search_dense(self, queries, k, flt=None):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        if len(self.live) == 0:
            return [(np.zeros(0, np.int64), np.zeros(0, np.float32)) for _ in range(q.shape[0])]
        sc = ocore.dense_scores(ocore.cosine_preprocess(q), self.x)
        mask = self._mask(flt).astype(np.uint8)
        return [ocore.topk(sc[i], k, mask) for i in range(q.shape[0])]

    def _live_sparse(self):
        return [r if self.live[i] else None for i, r in enumerate(self.sp)]

    def sparse_stats(self, ids):
        df, n = ocore.document_frequencies(self._live_sparse())
        return np.array([df.get(int(t), 0) for t in np.asarray(ids).reshape(-1)], np.int32), n

    def idf(self, n, df):
        return ocore.idf(n, df)

    def search_sparse(self, q_idx, q_val, k, flt=None, weights_given=False):
        rows = self._live_sparse()
        if len(rows) == 0 or len(np.atleast_1d(q_idx)) == 0:
            return np.zeros(0, np.int64), np.zeros(0, np.float32)
        qi, qv = np.asarray(q_idx, np.int32), np.asarray(q_val, np.float32)
        if weights_given:  # q_val already holds q_t * idf_t: ascending-id sum of f32 products, as the engine does it
            order = np.argsort(qi, kind="stable")
            qi, qv = qi[order], qv[order]
            sc = np.full(len(rows), -np.inf, np.float32)
            for r, row in enumerate(rows):
                if row is None:
                    continue
                acc, hit = np.float32(0), False
                for t, v in zip(row[0], row[1]):
                    j = np.searchsorted(qi, t)
                    if j < len(qi) and qi[j] == t:
                        acc = np.float32(acc + np.float32(qv[j] * v))
                        hit = True
                if hit:
                    sc[r] = acc
        else:
            sc = ocore.sparse_scores(rows, qi, qv, self.live)
        return ocore.topk(sc, k, self._mask(flt).astype(np.uint8))

    def search_hybrid(self, query, q_idx, q_val, limit, sparse_weight=0.1, fusion=0, flt=None):
        k = 3 * limit
        dr, ds = self.search_dense(np.asarray(query, np.float32).reshape(1, -1), k, flt)[0]
        sr, ss = self.search_sparse(q_idx, q_val, k, flt) if len(np.atleast_1d(q_idx)) else (np.zeros(0, np.int64), np.zeros(0, np.float32))
        fused = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit, sparse_weight, "json")
        return (np.array([r for r, _, _ in fused], np.int64), np.array([s for _, s, _ in fused], np.float64),
                np.array([int(f) for _, _, f in fused], np.int32))
