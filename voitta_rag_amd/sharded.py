"""Corpus sharded by document over the GPUs of one node: one process and one engine per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests) for the only
exchange step the path has — merging per-shard top-k lists.

The reference has no distributed code (one Qdrant server, vector_store.py:71); this module is
the multi-GPU form of VectorStoreService.search (:560-697):
  dense   every shard scans its own rows -> all_gather of k (global id, score) pairs -> merge
  sparse  IDF must be collection-wide: all_reduce(sum) of the query terms' document frequencies
          and of N, weights q_t * idf_t computed once (vr_idf), shards score with given weights
  hybrid  min-max fusion runs on the MERGED top-3*limit lists (:659-689), never per shard
Messages are k * 16 bytes per rank per list — latency-bound, so a hybrid query sends both of its lists in one
flat all_gather (after the all_reduce of the query terms' statistics).

Global row id of local row r on rank p: r * world + p (order-preserving per shard, unique).
Ties in score resolve to the lower global id.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch
import torch.distributed as dist

from .engine import fuse_minmax


def shard_of(file_path: str, world: int) -> int:
    """All chunks of one file live on one shard, so delete_by_file stays single-shard."""
    return zlib.crc32(file_path.encode("utf-8")) % world


class ShardedSearcher:
    def __init__(self, local, rank: int | None = None, world: int | None = None, group=None):
        """local: an Engine (or anything with search_dense / search_sparse / sparse_stats / idf)."""
        self.local = local
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        backend = dist.get_backend(group)
        self.comm_device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    # ---- id mapping ----------------------------------------------------------------------------
    def global_ids(self, rows: np.ndarray) -> np.ndarray:
        return rows.astype(np.int64) * self.world + self.rank

    def owner(self, gid: int) -> tuple[int, int]:
        return int(gid % self.world), int(gid // self.world)

    # ---- collectives ---------------------------------------------------------------------------
    def _merge(self, rows: np.ndarray, scores: np.ndarray, k: int):
        """all_gather this rank's (<= k) results, return the global top-k (gids, scores)."""
        return self._merge_lists([(rows, scores)], k)[0]

    def _merge_lists(self, lists, k: int):
        """One all_gather for several result lists of this rank (each <= k rows): returns the global
        top-k (gids, scores) of every list. The hybrid search sends its dense and its sparse list together."""
        n = len(lists)
        buf = np.full(n * 2 * k, -1, np.int64)
        for j, (rows, scores) in enumerate(lists):
            c = len(rows)
            base = j * 2 * k
            buf[base:base + c] = self.global_ids(rows)
            buf[base + k:base + k + c] = scores.astype(np.float32).view(np.int32).astype(np.int64)
        mine = torch.from_numpy(buf).to(self.comm_device)
        out = torch.empty(self.world * n * 2 * k, dtype=torch.int64, device=self.comm_device)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        allr = out.cpu().numpy().reshape(self.world, n, 2, k)
        merged = []
        for j in range(n):
            gids = allr[:, j, 0, :].reshape(-1)
            sc = allr[:, j, 1, :].reshape(-1).astype(np.int32).view(np.float32)
            keep = gids >= 0
            gids, sc = gids[keep], sc[keep]
            order = np.lexsort((gids, -sc.astype(np.float64)))[:k]  # score descending, then gid ascending
            merged.append((gids[order], sc[order]))
        return merged

    def global_sparse_weights(self, q_idx, q_val):
        """q_t * idf_t from all-reduced statistics -> (ids sorted unique, weights f32)."""
        pairs = {}
        for i, v in zip(np.asarray(q_idx, np.int64).tolist(), np.asarray(q_val, np.float32).tolist()):
            pairs.setdefault(int(i), np.float32(v))
        ids = np.array(sorted(pairs), np.int32)
        df, n_points = self.local.sparse_stats(ids)
        stat = torch.from_numpy(np.concatenate([df.astype(np.int64), [n_points]])).to(self.comm_device)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=self.group)
        stat = stat.cpu().numpy()
        n_all = int(stat[-1])
        w = np.array([np.float32(pairs[int(t)]) * np.float32(self.local.idf(n_all, int(d)))
                      for t, d in zip(ids, stat[:-1])], np.float32)
        return ids, w

    # ---- searches ------------------------------------------------------------------------------
    def search_dense(self, query, k: int, flt=None):
        rows, scores = self.local.search_dense(np.asarray(query, np.float32).reshape(1, -1), k, flt)[0]
        return self._merge(rows, scores, k)

    def search_sparse(self, q_idx, q_val, k: int, flt=None):
        ids, w = self.global_sparse_weights(q_idx, q_val)
        if len(ids) == 0:
            return np.zeros(0, np.int64), np.zeros(0, np.float32)
        rows, scores = self.local.search_sparse(ids, w, k, flt, weights_given=True)
        return self._merge(rows, scores, k)

    def search_hybrid(self, query, q_idx, q_val, limit: int, sparse_weight: float = 0.1, flt=None):
        """-> (gids, fused scores f64, from_dense) exactly as one engine holding every shard would.
        Two collectives per query: the all_reduce of the query terms' statistics, then ONE all_gather that
        carries this shard's dense and sparse lists together."""
        k = 3 * limit  # prefetch_limit, vector_store.py:636
        empty = (np.zeros(0, np.int64), np.zeros(0, np.float32))
        d_local = self.local.search_dense(np.asarray(query, np.float32).reshape(1, -1), k, flt)[0]
        s_local = empty
        have_sparse = len(np.atleast_1d(q_idx)) > 0
        if have_sparse:
            ids, w = self.global_sparse_weights(q_idx, q_val)
            if len(ids):
                s_local = self.local.search_sparse(ids, w, k, flt, weights_given=True)
        if have_sparse:
            (d_ids, d_sc), (s_ids, s_sc) = self._merge_lists([d_local, s_local], k)
        else:
            (d_ids, d_sc), (s_ids, s_sc) = self._merge_lists([d_local], k)[0], empty
        return fuse_minmax(d_ids, d_sc, s_ids, s_sc, limit, sparse_weight, True)
