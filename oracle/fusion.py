"""TEST INFRASTRUCTURE — Python restatement of VectorStoreService._hybrid_search's arithmetic,
reference: src/voitta/services/vector_store.py:634-697. Kept as close to the reference text as
possible: same names, same float operations, Python floats (f64).

Inputs are the two Qdrant result lists as [(id, score)], best first. ``transport="json"``
models how the reference receives scores: the server holds f32 scores and sends them over REST
as JSON; an f32 is printed with the shortest decimal that round-trips and Python parses that
decimal to the nearest f64 [EXT — qdrant-client default transport, pyproject.toml:31].
``transport="exact"`` widens f32 -> f64 exactly.

One deviation, documented in DESIGN.md: the reference iterates a Python ``set`` of ids
(:675-677) and sorts stably by score only (:689), so ties in the fused score come out in
arbitrary order. Here ids are visited in ascending order first, so ties resolve
to the lower id.
"""
import numpy as np


def _py_score(s, transport):
    if transport == "json":
        return float(str(np.float32(s)))
    return float(np.float32(s))


def hybrid_fuse(dense_results, sparse_results, limit, sparse_weight, transport="json"):
    dense_weight = 1.0 - sparse_weight  # :634

    def normalize(results):  # :659-669
        if not results:
            return {}
        scores = [_py_score(s, transport) for _, s in results]
        min_s, max_s = min(scores), max(scores)
        spread = max_s - min_s
        normed = {}
        for (rid, _), score in zip(results, scores):
            norm_score = (score - min_s) / spread if spread > 0 else 1.0
            normed[rid] = (norm_score, rid)
        return normed

    dense_normed = normalize(dense_results)
    sparse_normed = normalize(sparse_results)

    all_ids = sorted(set(dense_normed.keys()) | set(sparse_normed.keys()))  # :675 (+ order rule)
    combined = []
    for pid in all_ids:
        d_score = dense_normed[pid][0] if pid in dense_normed else 0.0
        s_score = sparse_normed[pid][0] if pid in sparse_normed else 0.0
        final_score = dense_weight * d_score + sparse_weight * s_score  # :680
        combined.append((final_score, pid, pid in dense_normed))

    combined.sort(key=lambda x: x[0], reverse=True)  # :689 (stable)
    return [(pid, score, from_dense) for score, pid, from_dense in combined[:limit]]  # :691-695


def rrf_fuse(dense_results, sparse_results, limit, k=2.0):
    """Qdrant-style reciprocal-rank fusion [EXT]; no reference code path uses it."""
    acc = {}
    for results in (dense_results, sparse_results):
        for pos, (rid, _) in enumerate(results):
            acc[rid] = acc.get(rid, 0.0) + 1.0 / (pos + k)
    dense_ids = {rid for rid, _ in dense_results}
    out = sorted(acc.items(), key=lambda kv: (-kv[1], kv[0]))[:limit]
    return [(rid, s, rid in dense_ids) for rid, s in out]
