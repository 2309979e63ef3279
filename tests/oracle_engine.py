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
        self.foreign_df: dict[int, int] = {}   # statistics of rows stored on other shards (df_apply)
        self.foreign_points = 0

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

    def _stats(self):
        df, n = ocore.document_frequencies(self._live_sparse())
        for t, c in self.foreign_df.items():
            df[t] = df.get(t, 0) + c
        return df, n + self.foreign_points

    def sparse_stats(self, ids):
        df, n = self._stats()
        return np.array([df.get(int(t), 0) for t in np.asarray(ids).reshape(-1)], np.int32), n

    def sparse_row_ids(self, rows, device=False):
        rows = np.asarray(rows, np.int64).reshape(-1)
        width = max([len(r[0]) for r in self.sp if r is not None] + [0])
        width = (width + 3) // 4 * 4
        out = np.full((len(rows), width), -1, np.int32)
        pts = 0
        for i, r in enumerate(rows):
            if self.live[r] and self.sp[r] is not None:
                out[i, : len(self.sp[r][0])] = self.sp[r][0]
                pts += 1
        return out, pts

    def df_apply(self, ids, n_points, sign=1):
        for t in np.asarray(ids).reshape(-1).tolist():
            if t >= 0:
                self.foreign_df[t] = self.foreign_df.get(t, 0) + sign
        self.foreign_points += sign * int(n_points)

    def idf(self, n, df):
        return ocore.idf(n, df)

    def search_sparse(self, q_idx, q_val, k, flt=None, weights_given=False):
        rows = self._live_sparse()
        if len(rows) == 0 or len(np.atleast_1d(q_idx)) == 0:
            return np.zeros(0, np.int64), np.zeros(0, np.float32)
        qi, qv = np.asarray(q_idx, np.int32).reshape(-1), np.asarray(q_val, np.float32).reshape(-1)
        first = {}
        for t, v in zip(qi.tolist(), qv.tolist()):
            first.setdefault(int(t), np.float32(v))   # a repeated id keeps its first value
        qi = np.array(sorted(first), np.int32)
        qv = np.array([first[int(t)] for t in qi], np.float32)
        if not weights_given:  # q_t * idf_t with the collection-wide statistic (own rows + those of other shards)
            df, n = self._stats()
            qv = np.array([np.float32(v) * np.float32(ocore.idf(n, df.get(int(t), 0))) for t, v in zip(qi, qv)], np.float32)
        # ascending-id sum of f32 products, as the engine does it
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
        return ocore.topk(sc, k, self._mask(flt).astype(np.uint8))

    def search_sparse_batch(self, sparse_queries, k, flt=None, weights_given=False):
        empty = (np.zeros(0, np.int64), np.zeros(0, np.float32))
        return [self.search_sparse(q[0], q[1], k, flt, weights_given) if q is not None and len(q[0]) else empty
                for q in sparse_queries]

    # ---- packed keys, merge (what the sharded searcher exchanges) -----------------------------------------------
    @staticmethod
    def _pack(rows, scores, k):
        out = np.zeros(k, np.uint64)
        if len(rows):
            u = np.ascontiguousarray(scores, np.float32).view(np.uint32)
            bits = np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint64)
            out[: len(rows)] = (bits << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.asarray(rows, np.uint64))
        return out

    def search_dense_keys(self, queries, k, flt=None, out=None):
        return np.stack([self._pack(r, s, k) for r, s in self.search_dense(queries, k, flt)])

    def search_hybrid_keys(self, queries, sparse_queries, k, flt=None, weights_given=False, out=None):
        dense = self.search_dense(queries, k, flt)
        sparse = self.search_sparse_batch(sparse_queries, k, flt, weights_given)
        return np.stack([np.stack([self._pack(*d, k), self._pack(*s, k)]) for d, s in zip(dense, sparse)])

    @staticmethod
    def merge_keys(parts, k):
        parts = np.asarray(parts).view(np.uint64)
        n_parts = parts.shape[0]
        lists = parts.reshape(n_parts, -1, k)
        n_lists = lists.shape[1]
        gid = np.full((n_lists, k), -1, np.int64)
        sc = np.zeros((n_lists, k), np.float32)
        cnt = np.zeros(n_lists, np.int32)
        for l in range(n_lists):
            cand = [(int(key), p) for p in range(n_parts) for key in lists[p, l] if key != 0]
            cand.sort(key=lambda kp: (-kp[0], kp[1]))
            for j, (key, p) in enumerate(cand[:k]):
                hi = np.uint32(key >> 32)
                u = np.uint32(hi ^ np.uint32(0x80000000)) if hi & np.uint32(0x80000000) else np.uint32(~hi)
                gid[l, j] = (0xFFFFFFFF - (key & 0xFFFFFFFF)) * n_parts + p
                sc[l, j] = np.array([u], np.uint32).view(np.float32)[0]
            cnt[l] = min(len(cand), k)
        return gid, sc, cnt

    def search_hybrid(self, query, q_idx, q_val, limit, sparse_weight=0.1, fusion=0, flt=None):
        k = 3 * limit
        dr, ds = self.search_dense(np.asarray(query, np.float32).reshape(1, -1), k, flt)[0]
        sr, ss = self.search_sparse(q_idx, q_val, k, flt) if len(np.atleast_1d(q_idx)) else (np.zeros(0, np.int64), np.zeros(0, np.float32))
        fused = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(sr.tolist(), ss.tolist())), limit, sparse_weight, "json")
        return (np.array([r for r, _, _ in fused], np.int64), np.array([s for _, s, _ in fused], np.float64),
                np.array([int(f) for _, _, f in fused], np.int32))

    def search_hybrid_batch(self, queries, sparse_queries, limit, sparse_weight=0.1, fusion=0, flt=None, raw=False):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        res = []
        for i in range(q.shape[0]):
            sq = sparse_queries[i]
            qi, qv = (sq[0], sq[1]) if sq is not None else ([], [])
            res.append(self.search_hybrid(q[i], qi, qv, limit, sparse_weight, fusion, flt))
        if not raw:
            return res
        nq = len(res)
        rows, scores = np.full((nq, limit), -1, np.int64), np.zeros((nq, limit), np.float64)
        fd, counts = np.zeros((nq, limit), np.int32), np.zeros(nq, np.int32)
        for i, (r, s, f) in enumerate(res):
            counts[i] = len(r)
            rows[i, : len(r)], scores[i, : len(r)], fd[i, : len(r)] = r, s, f
        return rows, scores, fd, counts
