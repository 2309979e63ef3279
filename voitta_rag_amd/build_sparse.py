"""Offline BM25 rebuild — the native form of the reference's ``scripts/build_sparse_vectors.py``
(``build_sparse_vectors``, :73-245). The reference scrolls the source
collection 500 points at a time, embeds ``payload["text"]`` with fastembed's BM25, and upserts
id + dense vector + payload (+ the sparse vector when the text is not empty, :158-194) into a
NEW collection ``<source>_v2`` created with ``Modifier.IDF`` (:44-70); the source is kept as a
backup and the operator switches over by setting ``QDRANT_COLLECTION`` (:236-243). It reports
``processed / elapsed`` chunks/sec (:218-221).

Here the "scroll" is ``vr_get_dense`` over the live rows, the BM25 model is the native tokenizer +
``vr_bm25_tf``, and the target collection is a second engine filled with ``vr_upsert`` (stored rows
are already cosine-preprocessed, so they go in bit for bit). Point ids, payloads, folder
dictionaries and timestamps carry over; rows come out compacted (tombstones are not copied). With
``VOITTA_INDEX_DIR`` set the target is saved there under its own name — a restart with
``QDRANT_COLLECTION=<target>`` then serves it, exactly the reference's switch-over — and
``switch=True`` additionally makes it the live index of this process (the source engine is closed).
"""
from __future__ import annotations

import logging
import time

import numpy as np

from . import bm25 as _bm25
from . import store_registry
from ._lib import VR_TS_ABSENT
from .config import get_settings
from .engine import Engine
from .vector_store import VectorStoreService, _Collection, get_vector_store

logger = logging.getLogger(__name__)


def build_sparse_vectors(batch_size: int = 500, insert_batch_size: int = 100, target_name: str | None = None,
                         dry_run: bool = False, switch: bool = False) -> dict:
    """Returns {"source", "target", "processed", "inserted", "skipped", "errors", "elapsed", "rate"}.
    ``insert_batch_size`` bounded HTTP payloads in the reference (:196-207) and has no meaning here."""
    settings = get_settings()
    source = settings.qdrant_collection
    target = target_name or f"{source}_v2"  # build_sparse_vectors.py:81
    vs = get_vector_store()
    engine = vs.client
    col = vs._col
    stats = {"source": source, "target": target, "processed": 0, "inserted": 0, "skipped": 0, "errors": 0,
             "elapsed": 0.0, "rate": 0.0}
    with col.write_lock, col.lock:  # (an offline tool: no mutation and no host-table change while it copies)
        rows = np.fromiter(col.live_rows(), dtype=np.int64)
        total_points = int(rows.size)
        logger.info("Source: %s  Target: %s  Points: %d  Dense dim: %d", source, target, total_points, vs.dimension)
        if total_points == 0:  # :100-102
            return stats
        t_engine = t_col = None
        if not dry_run:
            t_engine = Engine(vs.dimension, device=settings.gpu, initial_rows=total_points)
            t_col = _Collection()
            t_col.folder_ids = dict(col.folder_ids)
            t_col.index_folder_ids = dict(col.index_folder_ids)
        start = time.time()
        step = max(1, int(batch_size))
        for a in range(0, total_points, step):
            batch = rows[a:a + step]
            payloads = [col.payload[r] for r in batch]
            with_text = [i for i, p in enumerate(payloads) if p.get("text", "")]  # :158-165
            stats["skipped"] += len(batch) - len(with_text)
            stats["processed"] += len(batch)
            if dry_run:
                continue
            dense = engine.get_dense(batch)  # the stored (normalised) vectors, :144-150 with_vectors=True
            sparse = [None] * len(batch)
            if with_text:
                off, ids = _bm25.hashed_stems([payloads[i]["text"] for i in with_text])
                for i, row in zip(with_text, t_engine.bm25_tf(off, ids)):  # :168-170
                    sparse[i] = row
            # a point without text gets NO sparse vector (:176-185), which is not the same as an empty one
            # (it does not count towards the IDF's N): rows go in as runs with / without sparse vectors
            run_start = 0
            while run_start < len(batch):
                has = sparse[run_start] is not None
                run_end = run_start
                while run_end < len(batch) and (sparse[run_end] is not None) == has:
                    run_end += 1
                part = payloads[run_start:run_end]
                first = t_engine.upsert(
                    dense[run_start:run_end], sparse=sparse[run_start:run_end] if has else None,
                    folder_ids=np.array([t_col.folder_id(p["folder_path"], True) for p in part], np.int32),
                    index_folder_ids=np.array([t_col.index_folder_id(p.get("index_folder", p["folder_path"]), True)
                                               for p in part], np.int32),
                    created=np.array([VR_TS_ABSENT if p.get("source_created_at") is None else int(p["source_created_at"])
                                      for p in part], np.int64),
                    modified=np.array([VR_TS_ABSENT if p.get("source_modified_at") is None
                                       else int(p["source_modified_at"]) for p in part], np.int64))
                assert first == len(t_col.payload), "host table and engine rows diverged"
                for i in range(run_start, run_end):  # ids and payloads are kept (:187-193)
                    pid, payload = col.ids[batch[i]], payloads[i]
                    t_col.ids.append(pid)
                    t_col.payload.append(payload)
                    t_col.row_of[pid] = first + i - run_start
                    t_col.rows_by_file.setdefault(payload["file_path"], []).append(first + i - run_start)
                run_start = run_end
            stats["inserted"] += len(batch)
        if t_engine is not None:
            t_engine.sync()
        stats["elapsed"] = time.time() - start
        stats["rate"] = stats["processed"] / stats["elapsed"] if stats["elapsed"] > 0 else 0.0
        logger.info("Completed in %.1fs (%.0f chunks/sec); processed %d, inserted %d, skipped (no text) %d",
                    stats["elapsed"], stats["rate"], stats["processed"], stats["inserted"], stats["skipped"])  # :218-226
        if dry_run:
            return stats
        n_rows, n_live = t_engine.count()  # verification, :231-239
        if n_live != total_points:
            logger.warning("expected %d points in '%s', got %d", total_points, target, n_live)
        if settings.index_dir:
            saver = VectorStoreService()
            saver.collection_name = target
            saver._client = t_engine
            store_registry.collection(target, lambda: t_col)
            saver.save(settings.index_dir)
            store_registry.forget(target)
            logger.info("Saved '%s' under %s; to switch over set QDRANT_COLLECTION=%s", target, settings.index_dir, target)
        if switch:
            from . import embedding, sparse_embedding

            # the encoder lives in the engine it was loaded into: let the services bind to the new one
            embedding._embedding_service = None
            sparse_embedding._sparse_embedding_service = None
            store_registry.replace(t_engine, {source: t_col})
            vs._client = None
        elif not settings.index_dir:
            t_engine.close()
            logger.warning("neither VOITTA_INDEX_DIR nor switch=True: the rebuilt collection was discarded")
        else:
            t_engine.close()
    return stats


def main() -> None:  # python -m voitta_rag_amd.build_sparse [--batch-size N] [--target NAME] [--dry-run] [--switch]
    import argparse

    parser = argparse.ArgumentParser(description="Rebuild BM25 sparse vectors into a new native collection")
    parser.add_argument("--batch-size", type=int, default=500)
    parser.add_argument("--insert-batch-size", type=int, default=100)
    parser.add_argument("--target", default=None)
    parser.add_argument("--dry-run", action="store_true")
    parser.add_argument("--switch", action="store_true")
    args = parser.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    print(build_sparse_vectors(args.batch_size, args.insert_batch_size, args.target, args.dry_run, args.switch))


if __name__ == "__main__":
    main()
