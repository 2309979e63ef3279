"""Corpus sharded by document over the GPUs of one node: one process and one engine per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests) for the only
exchange step the path has — merging per-shard top-k lists.

The reference has no distributed code (one Qdrant server, vector_store.py:71); this module is
the multi-GPU form of VectorStoreService (:233-434 store / delete, :560-697 search, :699-1016 read helpers):
  dense   every shard scans its own rows -> all_gather of k (global id, score) pairs -> merge
  sparse  IDF must be collection-wide: all_reduce(sum) of the query terms' document frequencies
          and of N, weights q_t * idf_t computed once (vr_idf), shards score with given weights
  hybrid  min-max fusion runs on the MERGED top-3*limit lists (:659-689), never per shard
  batch   a whole query batch travels in ONE all_gather (and one all_reduce for the sparse statistics of all
          its queries); with the nccl backend the dense results never leave the device before the collective
          (Engine.search_dense_keys writes the ranking keys into the tensor RCCL sends)
  store   ShardedVectorStore: the service API, SPMD — every rank makes the same call with the same arguments;
          a chunk lives on rank shard_of(file_path), so store / delete_by_file / count_by_file touch one
          shard and the folder-level calls touch all of them and sum.
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


def _ordered_bits(scores: np.ndarray) -> np.ndarray:
    """f32 -> uint32 whose unsigned order is the float order (the high word of the engine's ranking keys)."""
    u = np.ascontiguousarray(scores, np.float32).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


class ShardedSearcher:
    def __init__(self, local, rank: int | None = None, world: int | None = None, group=None):
        """local: an Engine (or anything with search_dense / search_sparse / sparse_stats / idf)."""
        self.local = local
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        backend = dist.get_backend(group)
        self.on_device = backend == "nccl"
        self.comm_device = torch.device("cuda", torch.cuda.current_device()) if self.on_device else torch.device("cpu")

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

    def _merge_batch(self, gid: torch.Tensor, bits: torch.Tensor, k: int):
        """gid, bits: [nq, L, k] int64 on the communication device (-1 / 0 in empty slots): this rank's L result
        lists per query. ONE all_gather for the whole batch; returns (gids, order-preserving score bits), each
        [nq, L, k] on the host, merged over the ranks: score descending, then global id ascending."""
        nq, n_lists, _ = gid.shape
        mine = torch.stack([gid, bits]).contiguous()                       # [2, nq, L, k]
        flat = torch.empty(self.world * mine.numel(), dtype=torch.int64, device=self.comm_device)
        dist.all_gather_into_tensor(flat, mine.reshape(-1), group=self.group)
        out = flat.view((self.world,) + tuple(mine.shape))
        allg = out[:, 0].permute(1, 2, 0, 3).reshape(nq, n_lists, self.world * k)
        allb = out[:, 1].permute(1, 2, 0, 3).reshape(nq, n_lists, self.world * k)
        # empty slots last; then two stable sorts: by global id ascending, by score bits descending
        allg = torch.where(allg < 0, torch.full_like(allg, torch.iinfo(torch.int64).max), allg)
        order = torch.sort(allg, dim=-1, stable=True).indices
        allg, allb = torch.gather(allg, -1, order), torch.gather(allb, -1, order)
        order = torch.sort(allb, dim=-1, descending=True, stable=True).indices[..., :k]
        allg, allb = torch.gather(allg, -1, order), torch.gather(allb, -1, order)
        allg = torch.where(allb == 0, torch.full_like(allg, -1), allg)
        return allg.cpu().numpy(), allb.cpu().numpy()

    @staticmethod
    def _bits_to_scores(bits: np.ndarray) -> np.ndarray:
        u = bits.astype(np.uint32)
        return np.where(u & np.uint32(0x80000000), u ^ np.uint32(0x80000000), ~u).astype(np.uint32).view(np.float32)

    def global_sparse_weights(self, q_idx, q_val):
        """q_t * idf_t from all-reduced statistics -> (ids sorted unique, weights f32)."""
        return self.global_sparse_weights_batch([(q_idx, q_val)])[0]

    def global_sparse_weights_batch(self, queries):
        """The same for a batch of sparse queries with ONE all_reduce: the document frequencies of every query's
        terms and the point count N travel together."""
        uniq = []
        for q_idx, q_val in queries:
            pairs = {}
            for i, v in zip(np.asarray(q_idx, np.int64).reshape(-1).tolist(), np.asarray(q_val, np.float32).reshape(-1).tolist()):
                pairs.setdefault(int(i), np.float32(v))
            uniq.append(pairs)
        all_ids = np.array([t for pairs in uniq for t in sorted(pairs)], np.int32)
        df, n_points = self.local.sparse_stats(all_ids)
        stat = torch.from_numpy(np.concatenate([np.asarray(df, np.int64), [n_points]])).to(self.comm_device)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM, group=self.group)
        stat = stat.cpu().numpy()
        n_all = int(stat[-1])
        out, at = [], 0
        for pairs in uniq:
            ids = np.array(sorted(pairs), np.int32)
            w = np.array([np.float32(pairs[int(t)]) * np.float32(self.local.idf(n_all, int(d)))
                          for t, d in zip(ids, stat[at:at + len(ids)])], np.float32)
            at += len(ids)
            out.append((ids, w))
        return out

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

    # ---- batches: one all_gather per batch --------------------------------------------------------
    def _local_dense_batch(self, queries, k: int, flt):
        """-> (gid, bits) [nq, k] int64 tensors on the communication device for this shard's dense results."""
        nq = int(queries.shape[0])
        if self.on_device and hasattr(self.local, "search_dense_keys"):
            # the engine writes its ranking keys straight into the tensor the collective will read
            q = queries if hasattr(queries, "is_cuda") else torch.from_numpy(np.ascontiguousarray(queries, np.float32)).to(self.comm_device)
            keys = torch.empty((nq, k), dtype=torch.int64, device=self.comm_device)
            self.local.search_dense_keys(q.contiguous(), k, flt, out=keys)
            rows = 0xFFFFFFFF - (keys & 0xFFFFFFFF)
            bits = (keys >> 32) & 0xFFFFFFFF
            gid = torch.where(keys == 0, torch.full_like(rows, -1), rows * self.world + self.rank)
            return gid, torch.where(keys == 0, torch.zeros_like(bits), bits)
        res = self.local.search_dense(np.asarray(queries, np.float32).reshape(nq, -1), k, flt)
        gid = np.full((nq, k), -1, np.int64)
        bits = np.zeros((nq, k), np.int64)
        for i, (rows, scores) in enumerate(res):
            gid[i, :len(rows)] = self.global_ids(rows)
            bits[i, :len(rows)] = _ordered_bits(scores)
        return torch.from_numpy(gid).to(self.comm_device), torch.from_numpy(bits).to(self.comm_device)

    def search_dense_batch(self, queries, k: int, flt=None):
        """queries: (nq, D). -> list of (gids, scores) per query, merged over the shards; ONE all_gather."""
        gid, bits = self._local_dense_batch(queries, k, flt)
        g, b = self._merge_batch(gid[:, None, :], bits[:, None, :], k)
        out = []
        for i in range(g.shape[0]):
            keep = g[i, 0] >= 0
            out.append((g[i, 0][keep], self._bits_to_scores(b[i, 0][keep])))
        return out

    def search_hybrid_batch(self, queries, sparse_queries, limit: int, sparse_weight: float = 0.1, flt=None):
        """Hybrid search of a query batch: one all_reduce (statistics of every query's terms), one all_gather (every
        query's dense and sparse list). -> list of (gids, fused f64 scores, from_dense) per query."""
        k = 3 * limit
        nq = int(queries.shape[0])
        assert len(sparse_queries) == nq
        d_gid, d_bits = self._local_dense_batch(queries, k, flt)
        weights = self.global_sparse_weights_batch(sparse_queries)
        s_gid = np.full((nq, k), -1, np.int64)
        s_bits = np.zeros((nq, k), np.int64)
        for i, (ids, w) in enumerate(weights):
            if len(ids) == 0:
                continue
            rows, scores = self.local.search_sparse(ids, w, k, flt, weights_given=True)
            s_gid[i, :len(rows)] = self.global_ids(rows)
            s_bits[i, :len(rows)] = _ordered_bits(scores)
        gid = torch.stack([d_gid, torch.from_numpy(s_gid).to(self.comm_device)], dim=1)
        bits = torch.stack([d_bits, torch.from_numpy(s_bits).to(self.comm_device)], dim=1)
        g, b = self._merge_batch(gid, bits, k)
        out = []
        for i in range(nq):
            dk, sk = g[i, 0] >= 0, g[i, 1] >= 0
            out.append(fuse_minmax(g[i, 0][dk], self._bits_to_scores(b[i, 0][dk]), g[i, 1][sk],
                                   self._bits_to_scores(b[i, 1][sk]), limit, sparse_weight, True))
        return out


class ShardedVectorStore:
    """The VectorStoreService API over a corpus sharded by document, SPMD: every rank constructs one over its own
    local VectorStoreService and makes the SAME calls with the SAME arguments (like any torch.distributed
    collective); every rank gets the same return value — what one store holding every shard would return.

    Placement: a chunk lives on rank ``shard_of(metadata.file_path, world)``, so everything keyed by file
    (store_chunks of one file, delete_by_file, count_by_file, get_chunks_by_range, set_file_acl, page count)
    is one shard's work plus one small collective to share the answer; folder-level calls run on every
    shard and are summed or united. Point ids are generated once (by the owner) and shared, so a chunk's id is the
    same on every rank. Reference operations covered: vector_store.py:233-317 (store), :319-434 (deletes),
    :216-231 (ACL), :560-697 (search), :163-214 and :436-460 and :699-1016 (read helpers)."""

    def __init__(self, local_store, group=None):
        self.local = local_store
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.searcher = ShardedSearcher(local_store.client, self.rank, self.world, group)
        self.collection_name = local_store.collection_name
        self.dimension = local_store.dimension

    # ---- plumbing ---------------------------------------------------------------------------------
    def _gather(self, obj) -> list:
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def _sum(self, *values: int) -> list[int]:
        t = torch.tensor(list(values), dtype=torch.int64, device=self.searcher.comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.cpu().tolist()]

    def _mine(self, file_path: str) -> bool:
        return shard_of(file_path, self.world) == self.rank

    # ---- store / delete / ACL ----------------------------------------------------------------------
    def store_chunks(self, chunks, sparse_vectors=None, batch_size: int = 100) -> list[str]:
        """Every rank passes the full batch; each keeps the chunks of the files it owns. Returns the point ids in
        the order of ``chunks`` (vector_store.py:233-317), identical on every rank."""
        if not chunks:
            return []
        keep = [i for i, c in enumerate(chunks) if self._mine(c[2].file_path)]
        sv = None
        if sparse_vectors:
            sv = [sparse_vectors[i] if i < len(sparse_vectors) else ([], []) for i in keep]
        mine = self.local.store_chunks([chunks[i] for i in keep], sparse_vectors=sv, batch_size=batch_size) if keep else []
        ids: list = [None] * len(chunks)
        for part in self._gather(list(zip(keep, mine))):
            for i, pid in part:
                ids[i] = pid
        return ids

    def delete_by_file(self, file_path: str) -> int:
        return self._sum(self.local.delete_by_file(file_path) if self._mine(file_path) else 0)[0]

    def delete_by_folder(self, folder_path: str) -> int:
        return self._sum(self.local.delete_by_folder(folder_path))[0]

    def delete_by_index_folder(self, index_folder: str) -> int:
        return self._sum(self.local.delete_by_index_folder(index_folder))[0]

    def set_file_acl(self, file_path: str, allowed_users: list[str]) -> None:
        if self._mine(file_path):
            self.local.set_file_acl(file_path, allowed_users)
        dist.barrier(group=self.group)

    # ---- search ------------------------------------------------------------------------------------
    def search(self, query_embedding, limit: int = 10, folder_filter=None, include_folders=None, exclude_folders=None,
               exclude_index_folders=None, sparse_query=None, sparse_weight: float = 0.1, date_start=None,
               date_end=None, date_field=None):
        """VectorStoreService.search over every shard: the merged lists are fused once (vector_store.py:659-689),
        then the owners of the winners contribute their payloads (one more small collective)."""
        if limit <= 0:
            return []
        col = self.local._col
        q = np.asarray(query_embedding, dtype=np.float32).reshape(self.dimension)
        with col.lock:
            flt = self.local._build_filter(folder_filter, include_folders, exclude_folders, exclude_index_folders,
                                           date_start=date_start, date_end=date_end, date_field=date_field)
            hybrid = bool(sparse_query and self.local._has_sparse and len(sparse_query[0]) > 0)
            if hybrid:
                gids, scores, _ = self.searcher.search_hybrid(q, sparse_query[0], sparse_query[1], limit, sparse_weight, flt)
                scores = [float(s) for s in scores]
            else:
                gids, sc = self.searcher.search_dense(q, limit, flt)
                scores = [float(str(np.float32(s))) for s in sc]  # the REST/JSON transport of a dense score [EXT]
            mine = {}
            for gid in gids.tolist():
                p, row = self.searcher.owner(gid)
                if p == self.rank:
                    mine[gid] = (col.ids[row], col.payload[row])
        found = {}
        for part in self._gather(mine):
            found.update(part)
        return [self.local._chunk_from(found[g][0], found[g][1], s) for g, s in zip(gids.tolist(), scores)]

    # ---- read helpers --------------------------------------------------------------------------------
    def get_collection_info(self) -> dict:
        info = self.local.get_collection_info()
        n = self._sum(int(info.get("points_count", 0)))[0]
        return {"name": self.collection_name, "vectors_count": n, "points_count": n, "status": "green"}

    def count_by_file(self, file_path: str) -> int:
        return self._sum(self.local.count_by_file(file_path) if self._mine(file_path) else 0)[0]

    def count_chunks_for_files(self, file_paths: list[str]) -> dict[str, int]:
        out: dict[str, int] = {}
        for part in self._gather(self.local.count_chunks_for_files([f for f in file_paths if self._mine(f)])):
            out.update(part)
        return {f: out[f] for f in dict.fromkeys(file_paths) if f in out}

    def count_chunks_for_folder(self, folder_path: str) -> tuple[int, int]:
        files, chunks = self.local.count_chunks_for_folder(folder_path)
        files, chunks = self._sum(files, chunks)
        return files, chunks

    def get_folder_stats_batch(self, folder_paths: list[str]) -> dict[str, tuple[int, int]]:
        if not folder_paths:
            return {}
        local = self.local.get_folder_stats_batch(folder_paths)
        flat = self._sum(*[v for f in folder_paths for v in local.get(f, (0, 0))])
        return {f: (flat[2 * i], flat[2 * i + 1]) for i, f in enumerate(folder_paths)}

    def get_file_chunk_counts(self, folder_prefix: str = "") -> dict[str, int]:
        out: dict[str, int] = {}
        for part in self._gather(self.local.get_file_chunk_counts(folder_prefix)):
            out.update(part)
        return out

    def get_file_paths_by_index_folder(self, index_folder: str) -> set[str]:
        out: set[str] = set()
        for part in self._gather(self.local.get_file_paths_by_index_folder(index_folder)):
            out |= part
        return out

    def get_stored_page_count(self, file_path: str):
        vals = self._gather(self.local.get_stored_page_count(file_path) if self._mine(file_path) else None)
        return next((v for v in vals if v is not None), None)

    def get_chunks_by_range(self, file_path: str, first_chunk: int, last_chunk: int):
        parts = self._gather(self.local.get_chunks_by_range(file_path, first_chunk, last_chunk) if self._mine(file_path) else [])
        return [c for part in parts for c in part]

    def find_by_source_url(self, source_url: str):
        chunks = [c for part in self._gather(self.local.find_by_source_url(source_url)) for c in part]
        chunks.sort(key=lambda c: c.metadata.chunk_index)
        return chunks
