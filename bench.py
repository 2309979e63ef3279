#!/usr/bin/env python3
"""Contract benchmark of the voitta-rag hot path on MI355X.

Metric (BASELINE.json): chunks indexed/sec + p50 top-10 query latency @ 1M-chunk corpus, one GPU.
Workload (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU):
  1M-chunk corpus, bge-base-en-v1.5 shape (L12 H768 12 heads, CLS pooling) dense + BM25 sparse,
  hybrid top-10 (prefetch 30 per modality; the reference's min-max weighted fusion,
  vector_store.py:621-697 — SURVEY.md F3 explains why not RRF).

A STEP = one pass of the indexing hot path over one batch of synthetic chunks whose token ids are
already resident in HBM: dense encode (HIP BERT forward) -> BM25 token-count/TF weighting -> store
(cosine preprocess + MFMA-tiled corpus append + SELL sparse append + document frequencies) — the
three starred calls of IndexingService._index_file_standard (indexing.py:527-530,560) fused in
vr_index_batch. `value` = chunks/s over EXACTLY --steps such steps, max over ranks.
After the timed steps the query side is measured on the same engine (>= 1M rows per GPU):
single-stream hybrid top-10 queries, p50/p99 of the wall time per query (timed with the engine's event profiler off;
the scan kernels' timings come from a second pass over the same queries with it on).

Weak scaling: every rank owns its own 1M-row shard and indexes its own batches (no collective on
the indexing path); queries merge per-shard top-k lists with one RCCL all_gather per list.

Data: synthetic (seeded), weights random-init N(0, 0.02) of the named architecture — there is no
network for checkpoints. dtype: --precision f16 (default: f16 operands on the f16 MFMA, f32
accumulate, everything outside the matrix products in f32; |1 - cos| = 2e-6 against the f64 oracle
at the full 12 layers, north_star allows 1e-4), f16x3 (every operand carried as hi + lo f16 = 22 significant bits, three
f16-MFMA passes; |1 - cos| ~5e-8) or f32 (every product on the f32-input MFMA).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL = dict(name="bge-base-en-v1.5", layers=12, hidden=768, heads=12, intermediate=3072, vocab=30522,
             max_pos=512, pooling="cls")
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: Peak BF16/FP16 MFMA ~2.5 PF dense
PEAK_HBM_GBPS = 8000.0         # same guide: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy rate)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--batch", type=int, default=2200,
                   help="chunks per step per GPU (2200 x ~118 tokens = ~1014 row tiles of 256: the N=768 GEMMs then fill "
                        "11.9 of 12 rounds of the 256 CUs instead of 11.06 of 12 at 2048)")
    p.add_argument("--corpus", type=int, default=1_000_000, help="pre-populated rows per GPU")
    p.add_argument("--queries", type=int, default=1000)
    p.add_argument("--aniso-rows", type=int, default=1_000_000,
                   help="rows of the second, ANISOTROPIC corpus (common direction + noise, pairwise cosine 0.7) the dense search "
                        "is also measured on (0 = skip; single GPU only)")
    p.add_argument("--dropin-files", type=int, default=2000,
                   help="synthetic documents pushed through the reference's unmodified per-file call sequence (0 = skip)")
    p.add_argument("--other-rows", type=int, default=1_000_000,
                   help="rows of the dense corpora the OTHER model widths (MiniLM 384, bge-large 1024) are searched on, after a "
                        "short indexing run with each shape (0 = skip; single GPU only)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--precision", default="f16", choices=["f32", "f16x3", "f16"],
                   help="encoder matrix products: f16 operands / f32 accumulate (default; |1-cos| = 2e-6 vs the f64 "
                        "oracle at full depth, north_star allows 1e-4), (hi,lo) f16 split x3 passes (f32-class), or "
                        "the f32-input MFMA; tests/test_encoder_gpu.py holds all three")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    p.add_argument("--share-gpu", action="store_true",
                   help="rehearsal only: every rank uses GPU 0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    return p.parse_args()


def zipf_ids(gen, torch, n, vocab, dev, s=1.07):
    """Zipf(s)-like draw over [0, vocab) on the device (inverse-CDF of a continuous power law)."""
    u = torch.rand(n, device=dev, generator=gen)
    x = ((vocab ** (1.0 - s) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s))
    return (x - 1.0).clamp_(0, vocab - 1).to(torch.int64)


def stem_hash(torch, ids):
    """deterministic 31-bit id per synthetic stem (stands in for abs(murmur3(stem)))"""
    h = (ids * 2654435761 + 0x9E3779B9) & 0xFFFFFFFF
    h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
    return (h & 0x7FFFFFFF).to(torch.int32)


def make_batch(torch, gen, dev, n_chunks, seed_shift):
    """One step's input, device resident: WordPiece ids (~118 per chunk incl. [CLS]/[SEP], the length
    of the reference's 512-character chunks) and hashed BM25 stems (~48 per chunk)."""
    lens = torch.randint(96, 141, (n_chunks,), device=dev, generator=gen)
    wp_off = torch.zeros(n_chunks + 1, dtype=torch.int32, device=dev)
    wp_off[1:] = torch.cumsum(lens, 0).to(torch.int32)
    T = int(wp_off[-1])
    wp = (zipf_ids(gen, torch, T, MODEL["vocab"] - 1000, dev) + 999).to(torch.int32)
    wp[wp_off[:-1].long()] = 101            # [CLS]
    wp[(wp_off[1:] - 1).long()] = 102       # [SEP]
    blens = torch.randint(36, 61, (n_chunks,), device=dev, generator=gen)
    bm_off = torch.zeros(n_chunks + 1, dtype=torch.int64, device=dev)
    bm_off[1:] = torch.cumsum(blens, 0)
    bm = stem_hash(torch, zipf_ids(gen, torch, int(bm_off[-1]), 30000, dev))
    return wp.contiguous(), wp_off.contiguous(), bm.contiguous(), bm_off.contiguous(), T


def random_state(torch, gen, dev, model=None):
    from voitta_rag_amd import encoder as enc

    model = model or MODEL
    H, I = model["hidden"], model["intermediate"]
    state = {}
    for n in enc.tensor_names(model["layers"]):
        if n.endswith("word_embeddings.weight"): shp = (model["vocab"], H)
        elif n.endswith("position_embeddings.weight"): shp = (model["max_pos"], H)
        elif n.endswith("token_type_embeddings.weight"): shp = (2, H)
        elif n.endswith("intermediate.dense.weight"): shp = (I, H)
        elif n.endswith("intermediate.dense.bias"): shp = (I,)
        elif n.endswith("output.dense.weight") and "attention" not in n: shp = (H, I)
        elif n.endswith(".weight") and "LayerNorm" not in n: shp = (H, H)
        else: shp = (H,)
        t = torch.randn(shp, device=dev, generator=gen) * 0.02
        if "LayerNorm.weight" in n:
            t = t + 1.0
        state[n] = t
    return state


def populate(torch, gen, dev, engine, rows, dim):
    """Search corpus generated directly on the device as seeded unit vectors + synthetic BM25 rows
    (SURVEY.md §8d allows this for the search side). Returns the dense tensor for the recall check."""
    keep = []
    chunk = 100_000
    nnz = 40
    for a in range(0, rows, chunk):
        n = min(chunk, rows - a)
        x = torch.nn.functional.normalize(torch.randn((n, dim), device=dev, generator=gen), dim=1).contiguous()
        # BM25 rows over the SAME vocabulary the queries and the timed indexing steps draw from (Zipf(1.07)
        # over 30,000 words; term id = stem_hash(word)): 40 words per row, a word drawn twice is replaced by a rare one
        # (ids >= 30000: the long tail every real collection has). Until round 2 the row ids were stem_hash(word * 64 +
        # position) — a different id space from the queries' stem_hash(word), so most query terms named no row at all.
        words = zipf_ids(gen, torch, n * nnz, 30000, dev).view(n, nnz)
        words, _ = torch.sort(words, dim=1)
        dup = torch.zeros_like(words, dtype=torch.bool)
        dup[:, 1:] = words[:, 1:] == words[:, :-1]
        rare = 30000 + torch.randint(0, 1 << 22, (n, nnz), device=dev, generator=gen)
        words = torch.where(dup, rare, words)
        ids, _ = torch.sort(stem_hash(torch, words).to(torch.int64), dim=1)
        dup = torch.zeros_like(ids, dtype=torch.bool)
        dup[:, 1:] = ids[:, 1:] == ids[:, :-1]
        ids = torch.where(dup, ids + 1, ids)          # (rare) hash collisions inside a row
        ids, _ = torch.sort(ids, dim=1)
        off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * nnz).contiguous()
        val = (torch.rand(n * nnz, device=dev, generator=gen) * 1.6 + 0.4).contiguous()
        engine.upsert(x, sparse=(off, ids.to(torch.int32).reshape(-1).contiguous(), val))
        keep.append(x)
    return keep


def synthetic_vocab(rng, size):
    """A WordPiece vocabulary of exactly ``size`` entries: specials, single characters, '##' characters, random words."""
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789.,!?:;'-")
    vocab += ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    words: dict = {}
    while len(vocab) + len(words) < size:
        for w in ("".join(rng.choice(letters, size=int(rng.integers(2, 8)))) for _ in range(4096)):
            if len(vocab) + len(words) < size and w not in words and len(w) > 1:
                words[w] = True
    words = list(words)
    # [CLS] / [SEP] must sit where the bench's device-made batches put them (ids 101 / 102), as in BERT's vocab.txt
    vocab = vocab + words
    for tok, at in (("[CLS]", 101), ("[SEP]", 102)):
        i = vocab.index(tok)
        vocab[i], vocab[at] = vocab[at], vocab[i]
    return vocab, words


def synthetic_documents(rng, words, n_files):
    """Parsed files as IndexingService sees them: 3..40 paragraphs of 1..4 sentences of 5..24 words."""
    warr = np.array(words)
    docs = []
    for _ in range(n_files):
        paras = []
        for _ in range(int(rng.integers(3, 40))):
            n_sent = int(rng.integers(1, 5))
            lens = rng.integers(5, 25, size=n_sent)
            ws = warr[rng.integers(0, len(warr), size=int(lens.sum()))]
            sents, a = [], 0
            for ln in lens:
                sents.append(" ".join(ws[a:a + ln]).capitalize() + ".")
                a += ln
            paras.append(" ".join(sents))
        docs.append("\n\n".join(paras))
    return docs


def mirror_engine_rows(vs, n_rows):
    """Host-table rows for engine rows that were appended BELOW the service (the device-made corpus and the timed
    index_batch steps), so that VectorStoreService can resolve any row a search returns."""
    col = vs._col
    have = len(col.ids)
    if n_rows > have:
        payload = vs._payload_of("synthetic corpus row", _placeholder_meta())
        vs._append_host_rows(col, [f"synthetic-{i}" for i in range(have, n_rows)], [payload] * (n_rows - have))


def _placeholder_meta():
    from voitta_rag_amd.vector_store import ChunkMetadata

    return ChunkMetadata(file_path="synthetic/corpus.bin", folder_path="synthetic", index_folder="synthetic",
                         file_name="corpus.bin", chunk_index=0, total_chunks=1, start_char=0, end_char=0, indexed_at="t")


def dropin_section(args, engine, state, rng):
    """The reference's own caller code on the native classes, unmodified:
      indexing  IndexingService._index_file_standard (indexing.py:513-563): chunk_text -> embed_texts -> sparse
                embed_texts -> zip -> store_chunks, one file at a time, from raw text;
      query     mcp_server.py:469-485: embed_query(text) -> sparse embed_query(text) -> VectorStoreService.search
    on the engine that holds the >= 1M-row corpus. Returns (chunks/s, chunks, files, p50 ms, p99 ms)."""
    import tempfile

    from voitta_rag_amd import config, embedding, sparse_embedding, store_registry, vector_store
    from voitta_rag_amd import encoder as enc
    from voitta_rag_amd.chunking import get_chunking_service
    from voitta_rag_amd.vector_store import ChunkMetadata
    from voitta_rag_amd.wordpiece import WordPieceTokenizer

    vocab, words = synthetic_vocab(rng, MODEL["vocab"])
    d = tempfile.mkdtemp(prefix="voitta-bench-")
    with open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    os.environ["EMBEDDING_DIMENSION"] = str(MODEL["hidden"])
    os.environ["EMBEDDING_MODEL"] = MODEL["name"]  # (no "e5" in the name: no prefixes, as for bge)
    config.get_settings.cache_clear()
    store_registry.set_engine(engine)
    desc = enc.BertDesc(MODEL["layers"], MODEL["hidden"], MODEL["heads"], MODEL["intermediate"], vocab=MODEL["vocab"],
                        max_pos=MODEL["max_pos"], pooling=MODEL["pooling"], normalize=True, precision=args.precision)
    emb = embedding.EmbeddingService()
    emb._model = embedding.NativeSentenceEncoder(engine, desc, state, WordPieceTokenizer.from_pretrained(d), MODEL["max_pos"])
    embedding._embedding_service = emb
    sp = sparse_embedding.get_sparse_embedding_service()
    vs = vector_store.VectorStoreService()
    mirror_engine_rows(vs, engine.count()[0])
    chunker = get_chunking_service()
    docs = synthetic_documents(rng, words, args.dropin_files + 40)

    def index_file(i, content):
        fp_ = f"dir{i % 7}/f{i}.md"
        vs.count_by_file(fp_)    # the skip check of every file (indexing.py:239)
        vs.delete_by_file(fp_)   # "delete existing chunks BEFORE parsing", every file, new ones included (indexing.py:281-288)
        chunks = chunker.chunk_text(content)
        texts = [c.text for c in chunks]
        embeddings = emb.embed_texts(texts)
        sparse_vectors = sp.embed_texts(texts)
        fp = f"dir{i % 7}/f{i}.md"
        chunk_data = [(c.text, e, ChunkMetadata(file_path=fp, folder_path=f"dir{i % 7}", index_folder=f"dir{i % 7}",
                                                file_name=f"f{i}.md", chunk_index=c.index, total_chunks=len(chunks),
                                                start_char=c.start_char, end_char=c.end_char, indexed_at="t",
                                                source_modified_at=1_700_000_000 + i))
                      for c, e in zip(chunks, embeddings)]
        vs.store_chunks(chunk_data, sparse_vectors=sparse_vectors)
        return len(chunks)

    # default mode of the drop-in (VOITTA_DEFERRED_INDEXING unset): store_chunks makes its one fused call itself and
    # fails where the reference's upsert fails; timed on a tenth of the files
    os.environ.pop("VOITTA_DEFERRED_INDEXING", None)
    n_sync = max(20, args.dropin_files // 10)
    for i in range(20):
        index_file(10_000_000 + i, docs[i])
    engine.sync()
    t0 = time.perf_counter()
    sync_chunks = sum(index_file(20_000_000 + i, docs[40 + i]) for i in range(n_sync))
    engine.sync()
    sync_rate = sync_chunks / (time.perf_counter() - t0)
    # the opt-in write-behind (VOITTA_DEFERRED_INDEXING=1: a caller that calls flush() before it commits its bookkeeping)
    os.environ["VOITTA_DEFERRED_INDEXING"] = "1"
    for i in range(40):  # warm-up: flusher thread, table growth, graphs
        index_file(i, docs[i])
    vs.flush()
    engine.sync()
    t0 = time.perf_counter()
    n_chunks = sum(index_file(40 + i, docs[40 + i]) for i in range(args.dropin_files))
    vs.flush()  # every row searchable
    engine.sync()
    dt = time.perf_counter() - t0
    assert vs.failed_file_paths() == []
    os.environ.pop("VOITTA_DEFERRED_INDEXING", None)

    warr = np.array(words)
    questions = [" ".join(warr[rng.integers(0, len(warr), size=int(rng.integers(5, 12)))]) + "?" for _ in range(220)]
    lat = np.empty(200)
    for i, q in enumerate(questions):
        t1 = time.perf_counter()
        got = vs.search(emb.embed_query(q), limit=10, sparse_query=sp.embed_query(q), sparse_weight=0.1)
        if i >= 20:
            lat[i - 20] = time.perf_counter() - t1
        assert len(got) == 10
    if os.environ.get("VR_BENCH_TEXT_DEBUG"):  # where a question's time goes in THIS process state (stderr; not part of the line)
        tok = emb.model.tokenizer._h
        l2 = []
        for q in questions[20:120]:
            t1 = time.perf_counter()
            engine.query_text(tok, q, q, 512, 10, 0.1)
            l2.append(time.perf_counter() - t1)
        ids, off = emb.model.tokenize([questions[50]])
        import torch as _t
        out = _t.empty((1, engine.dim), device=ids.device)
        l3 = []
        for _ in range(100):
            t1 = time.perf_counter()
            enc.encode(engine, ids, off, out)
            engine.sync()
            l3.append(time.perf_counter() - t1)
        print(f"[text debug] from text p50 {np.percentile(lat, 50) * 1e3:.4f} ms; engine.query_text alone {np.percentile(l2, 50) * 1e3:.4f}; "
              f"encode alone {np.percentile(l3, 50) * 1e3:.4f}; rows {engine.count()}", file=sys.stderr)
    store_registry.set_engine(None)  # (the engine stays ours to close)
    return (n_chunks / dt, n_chunks, args.dropin_files, float(np.percentile(lat, 50) * 1e3), float(np.percentile(lat, 99) * 1e3),
            sync_rate, n_sync)


def anisotropic_section(args, torch, dev, dim):
    """Dense top-10 on rows that share a common direction (pairwise cosine 0.7, what sentence-embedding collections
    look like): scores sit three times closer together than on random unit rows while the int8 bound is as wide, which
    is what the centred shadow (csrc/prefilter.hip) is for. Queries are drawn like the rows."""
    from voitta_rag_amd import Engine

    gen = torch.Generator(device=dev).manual_seed(4242)
    common = torch.nn.functional.normalize(torch.randn(dim, device=dev, generator=gen), dim=0)

    def make(n):
        u = torch.randn((n, dim), device=dev, generator=gen)
        u = torch.nn.functional.normalize(u - (u @ common)[:, None] * common[None, :], dim=1)
        return torch.nn.functional.normalize((0.7 ** 0.5) * common[None, :] + (0.3 ** 0.5) * u, dim=1).contiguous()

    e = Engine(dim, device=dev.index or 0, initial_rows=args.aniso_rows)
    for a in range(0, args.aniso_rows, 100_000):
        e.upsert(make(min(100_000, args.aniso_rows - a)))
    qs = make(220).cpu().numpy()
    for i in range(20):
        e.search_dense(qs[i:i + 1], 10)
    s0 = e.stats()
    lat, cand = np.empty(200), np.empty(200)
    for i in range(200):
        t0 = time.perf_counter()
        e.search_dense(qs[20 + i:21 + i], 10)
        lat[i] = time.perf_counter() - t0
        cand[i] = e.stats()["last_candidates"]
    s1 = e.stats()
    qb = make(args.queries).cpu().numpy()
    e.search_dense(qb, 10)
    t0 = time.perf_counter()
    for _ in range(3):
        e.search_dense(qb, 10)
    dt = (time.perf_counter() - t0) / 3
    s2 = e.stats()
    out = {"rows": args.aniso_rows, "pairwise_cosine": 0.7,
           "p50_dense_top10_ms": round(float(np.percentile(lat, 50) * 1e3), 4),
           "p99_dense_top10_ms": round(float(np.percentile(lat, 99) * 1e3), 4),
           "candidates_median": int(np.median(cand)), "candidates_max": int(cand.max()),
           "fallbacks": s1["fallback"] - s0["fallback"],
           "qps_batched": round(args.queries / dt, 1), "batched_fallbacks": s2["batch_fallback"] - s1["batch_fallback"]}
    e.close()
    torch.cuda.empty_cache()
    return out


OTHER_MODELS = [
    dict(name="all-MiniLM-L6-v2", layers=6, hidden=384, heads=12, intermediate=1536, vocab=30522, max_pos=512, pooling="mean",
         config="BASELINE configs[1] / configs[3] width"),
    dict(name="bge-large-en-v1.5", layers=24, hidden=1024, heads=16, intermediate=4096, vocab=30522, max_pos=512, pooling="cls",
         config="BASELINE configs[4] width"),
]


def other_width_section(args, torch, dev, model, rows):
    """The other model widths of BASELINE.json, driver-timed on a short run (outside `value`): the same fused indexing
    step (encode + BM25 tf + store) with the named shape, and the dense / batched searches on a corpus of `rows` unit
    rows of that width."""
    from voitta_rag_amd import Engine
    from voitta_rag_amd import encoder as enc

    dim = model["hidden"]
    gen = torch.Generator(device=dev).manual_seed(77)
    steps, warm = 3, 1
    e = Engine(dim, device=dev.index or 0, initial_rows=rows + (steps + warm) * args.batch + 64)
    state = random_state(torch, gen, dev, model)
    enc.load_encoder(e, enc.BertDesc(model["layers"], dim, model["heads"], model["intermediate"], vocab=model["vocab"],
                                     max_pos=model["max_pos"], pooling=model["pooling"], precision=args.precision), state)
    del state
    batches = [make_batch(torch, gen, dev, args.batch, i) for i in range(steps + warm)]
    for b in batches[:warm]:
        e.index_batch(b[0], b[1], b[2], b[3])
    e.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[warm:]:
        e.index_batch(b[0], b[1], b[2], b[3])
    e.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_n, gemm_flop = e.profile_read(Engine.PROF_GEMM)
    e.profile(False)
    del batches
    for a in range(0, rows, 100_000):
        n = min(100_000, rows - a)
        e.upsert(torch.nn.functional.normalize(torch.randn((n, dim), device=dev, generator=gen), dim=1).contiguous())
    qs = torch.nn.functional.normalize(torch.randn((args.queries + 20, dim), device=dev, generator=gen), dim=1).cpu().numpy()
    for i in range(20):
        e.search_dense(qs[i:i + 1], 10)
    lat = np.empty(200)
    for i in range(200):
        t1 = time.perf_counter()
        e.search_dense(qs[20 + i:21 + i], 10)
        lat[i] = time.perf_counter() - t1
    qb = np.ascontiguousarray(qs[20:20 + args.queries])
    e.search_dense(qb, 10)
    t1 = time.perf_counter()
    for _ in range(3):
        e.search_dense(qb, 10)
    bdt = (time.perf_counter() - t1) / 3
    gemm_tf = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    out = {"model": model["name"], "shape": f"L{model['layers']} H{dim} {model['pooling']}", "stands_for": model["config"],
           "chunks_per_s": round(steps * args.batch / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "gemm_TFLOPs": round(gemm_tf, 1), "gemm_frac_of_f16_peak": round(gemm_tf / PEAK_F16_MFMA_TFLOPS, 4),
           "gemm_share_of_step_time": round(gemm_ms * 1e-3 / dt, 4),
           "corpus_rows": rows, "p50_dense_top10_ms": round(float(np.percentile(lat, 50) * 1e3), 4),
           "p99_dense_top10_ms": round(float(np.percentile(lat, 99) * 1e3), 4),
           "qps_dense_batched": round(args.queries / bdt, 1)}
    e.close()
    torch.cuda.empty_cache()
    return out


def pmc_traffic(kernel_prefix: str):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, see profiles/README.md), corrected
    as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE reports half
    the bytes of a wide coalesced read, so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
    None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_fetch_write_summary.json")
    if not os.path.exists(path):
        return None
    try:
        summ = json.load(open(path))
        n = f = w = 0.0
        for name, row in summ.items():
            if kernel_prefix in name:
                n += row["launches"]
                f += row["FETCH_SIZE_KB_avg"] * row["launches"]
                w += row["WRITE_SIZE_KB_avg"] * row["launches"]
        return None if n == 0 else round((2.0 * f + w) * 1024.0 / n)
    except Exception:
        return None


def host_cores() -> int:
    """Cores this process may actually use: cgroup quota and affinity, not the machine's count
    (a 1-GPU box exposes 256 logical CPUs but schedules a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("VR_BENCH_CPU_THREADS")
    return int(env) if env else min(n, 16)  # 16 = the CPU share of one GPU on the bench box


def cpu_baseline(args, rng):
    """The oracle timed on this box's host cores on a bounded sample of the same workload."""
    import torch

    from oracle import bert as obert
    from oracle import bert_torch
    from oracle import bm25 as obm
    from oracle import core as ocore

    torch.set_num_threads(host_cores())
    shape = obert.BertShape(MODEL["layers"], MODEL["hidden"], MODEL["heads"], MODEL["intermediate"],
                            vocab=MODEL["vocab"], max_pos=MODEL["max_pos"])
    model = bert_torch.TorchBert(obert.random_weights(shape, 7), shape, MODEL["pooling"])
    model.encode([rng.integers(999, MODEL["vocab"], size=16).astype(np.int32) for _ in range(4)])  # thread-pool warm-up
    done, t_used = 0, 0.0
    while t_used < args.cpu_seconds and done < 4096:
        seqs = [rng.integers(999, MODEL["vocab"], size=int(rng.integers(96, 141))).astype(np.int32) for _ in range(32)]
        stems = [rng.integers(0, 1 << 31, size=int(rng.integers(36, 61))).tolist() for _ in range(32)]
        t0 = time.perf_counter()
        emb = model.encode(seqs, batch_size=32)           # embed_texts
        sp = [obm.tf_from_hashed(s) for s in stems]       # sparse embed_texts (token-count / tf)
        ocore.cosine_preprocess(emb)                      # store: Qdrant-side normalisation
        t_used += time.perf_counter() - t0
        done += 32
        del sp
    chunks_per_s = done / t_used
    # query side: exact f32 brute force (torch matvec + top-30 on the same threads) over a host
    # matrix of the full corpus size, so that it streams from DRAM like the real thing would
    rows = args.corpus
    block = rng.standard_normal((50_000, MODEL["hidden"]), dtype=np.float32)
    block /= np.linalg.norm(block, axis=1, keepdims=True)
    x = torch.from_numpy(block).repeat((rows + 49_999) // 50_000, 1)[:rows].contiguous()
    lat = []
    with torch.no_grad():
        for _ in range(8):
            q = torch.from_numpy(rng.standard_normal(MODEL["hidden"], dtype=np.float32))
            t0 = time.perf_counter()
            torch.topk(torch.mv(x, q), 30)
            lat.append(time.perf_counter() - t0)
    q_ms = float(np.median(lat[2:]) * 1e3)
    del x
    return {
        "value": round(chunks_per_s, 2), "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{done} chunks (batches of 32, 96-140 tokens) through oracle/bert_torch.py (torch-CPU f32, "
                  f"{MODEL['name']} shape, all host threads) + oracle BM25 tf + oracle cosine preprocess, {t_used:.1f} s",
        "query_p50_ms": round(q_ms, 2),
        "query_sample": f"median of 6 exact f32 dense scans (torch.mv + topk(30), {torch.get_num_threads()} threads) "
                        f"over {rows} x {MODEL['hidden']} host rows; sparse scan and fusion not included",
    }


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, the same
    command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment — what torch.distributed.run would
    set) and wait for them. This parent never touches the GPU (nothing here initialises HIP; a process that has must
    not be replaced, and is not: the ranks are children). Rank 0 prints the JSON line on the inherited stdout. Any
    rank failing ends the others and makes the exit code non-zero."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = set(range(len(procs)))
    while pending:
        for i in list(pending):
            code = procs[i].poll()
            if code is None:
                continue
            pending.discard(i)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {i} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for j in pending:
                    procs[j].terminate()
        time.sleep(0.05)
    return rc if rc >= 0 else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N, or plain "
                         "python bench.py --gpus N, which starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from voitta_rag_amd import Engine
    from voitta_rag_amd import encoder as enc

    dim = MODEL["hidden"]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    engine = Engine(dim, device=local_rank, initial_rows=args.corpus + (args.steps + args.warmup) * args.batch + (args.dropin_files + 40) * 32 + 64)
    state = random_state(torch, gen, dev)
    enc.load_encoder(engine, enc.BertDesc(MODEL["layers"], dim, MODEL["heads"], MODEL["intermediate"],
                                          vocab=MODEL["vocab"], max_pos=MODEL["max_pos"], pooling=MODEL["pooling"],
                                          precision=args.precision), state)
    corpus_chunks = populate(torch, gen, dev, engine, args.corpus, dim)
    batches = [make_batch(torch, gen, dev, args.batch, i) for i in range(args.warmup + args.steps)]
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the corpus is sharded by document, every rank indexes its own batches. The only exchange of the indexing
    # path is the one that keeps the BM25 document frequencies collection-wide: after a batch is stored
    # its term ids are all-gathered and applied to the other shards' tables — INSIDE the timed step.
    searcher = None
    if world > 1:
        from voitta_rag_amd.sharded import ShardedSearcher

        searcher = ShardedSearcher(engine)
        searcher.replicate_all()  # the pre-populated shards were filled locally

    def index_step(b):
        first = engine.index_batch(b[0], b[1], b[2], b[3])
        if searcher is not None:
            searcher.rows_added(np.arange(first, first + args.batch, dtype=np.int64))

    # ---- indexing: W warm-up steps, then exactly K timed steps ------------------------------------
    for b in batches[: args.warmup]:
        index_step(b)
    engine.profile(True)
    barrier()
    t0 = time.perf_counter()
    tokens = 0
    for b in batches[args.warmup:]:
        index_step(b)
        tokens += b[4]
    barrier()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_n, gemm_flop = engine.profile_read(Engine.PROF_GEMM)
    attn_ms, attn_n, attn_flop = engine.profile_read(Engine.PROF_ATTENTION)
    attn_bytes = MODEL["layers"] * tokens * 4 * dim * 2.0  # per layer: Q, K, V read + context written, f16
    engine.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    chunks_per_s = world * args.steps * args.batch / dt

    # ---- queries: single stream, hybrid top-10 over >= corpus rows per GPU ------------------------
    n_rows, _ = engine.count()
    qgen = torch.Generator(device=dev).manual_seed(99)   # same queries on every rank
    qs = torch.nn.functional.normalize(torch.randn((args.queries + 20, dim), device=dev, generator=qgen), dim=1)
    qs_host = qs.cpu().numpy()
    q_terms = stem_hash(torch, zipf_ids(qgen, torch, (args.queries + 20) * 6, 30000, dev)).view(-1, 6).cpu().numpy()
    q_nnz = np.random.default_rng(5).integers(4, 7, size=args.queries + 20)
    ones = np.ones(8, np.float32)
    if world > 1:
        search = lambda i: searcher.search_hybrid(qs_host[i], q_terms[i, : q_nnz[i]], ones[: q_nnz[i]], 10, 0.1)  # noqa: E731
    else:
        search = lambda i: engine.search_hybrid(qs_host[i], q_terms[i, : q_nnz[i]], ones[: q_nnz[i]], 10, 0.1)  # noqa: E731
    for i in range(20):
        search(i)
    barrier()
    lat = np.empty(args.queries)
    for i in range(args.queries):  # the latencies: the engine's event profiler is OFF (its four event records per query
        t1 = time.perf_counter()   # cost ~10 us of the wall time they would be part of)
        search(20 + i)
        lat[i] = time.perf_counter() - t1
    barrier()
    engine.profile(True)           # the kernel timings of the same queries: a second pass with the profiler on
    for i in range(min(args.queries, 300)):
        search(20 + i)
    barrier()
    scan_ms, scan_n, scan_bytes = engine.profile_read(Engine.PROF_DENSE_SCAN)
    sp_ms, sp_n, sp_bytes = engine.profile_read(Engine.PROF_SPARSE_SCAN)
    engine.profile(False)
    p50, p99 = float(np.percentile(lat, 50) * 1e3), float(np.percentile(lat, 99) * 1e3)
    if world > 1:
        t = torch.tensor([p50, p99], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        p50, p99 = float(t[0]), float(t[1])
    dense_only = [engine.search_dense(qs_host[20 + i: 21 + i], 10) for i in range(min(50, args.queries))]

    # ---- batched queries: all --queries dense top-10 searches in ONE call -----------
    # host arrays in, host arrays out (queries up, rows + scores down are inside the timed region)
    qb = np.ascontiguousarray(qs_host[20:20 + args.queries])
    # (N = 1: the result arrays as the C-ABI fills them — rows / scores [queries, 10] and counts; cutting them into a Python
    # list of 2000 small arrays is a tenth of the call)
    batch_call = (lambda: searcher.search_dense_batch(qb, 10)) if world > 1 else (lambda: engine.search_dense(qb, 10, raw=True))
    batch_reps = 5
    try:
        for _ in range(2):
            batched = batch_call()
        engine.profile(True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(batch_reps):
            batched = batch_call()
        barrier()
        batch_dt = time.perf_counter() - t1
    except Exception as exc:  # the batched section is outside `value`: a failure here must not cost the contract line
        print(f"[bench] batched-query section failed on rank {rank}: {exc!r}", file=sys.stderr, flush=True)
        batched, batch_dt = None, float("nan")
    bscan_ms, bscan_n, bscan_ops = engine.profile_read(Engine.PROF_BATCH_SCAN)
    engine.profile(False)
    if world > 1:
        t = torch.tensor([batch_dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        batch_dt = float(t.item())
    qps_batched = batch_reps * args.queries / batch_dt if batch_dt == batch_dt else None
    # ---- configs[4]'s query side: all --queries HYBRID top-10 searches in ONE call (dense batch + sparse batch beside
    # it + fusion on the host threads; N > 1: one all_gather for the whole batch, merge on the engine) ---------------
    sq_list = [(q_terms[20 + i, : q_nnz[20 + i]], ones[: q_nnz[20 + i]]) for i in range(args.queries)]
    if world == 1:  # the batch's sparse queries as ONE CSR triple (offsets, ids, values): host arrays in, host arrays out
        sq_off = np.zeros(args.queries + 1, np.int64)
        sq_off[1:] = np.cumsum(q_nnz[20:20 + args.queries])
        sq_list = (sq_off, np.concatenate([q[0] for q in sq_list]).astype(np.int32), np.ones(int(sq_off[-1]), np.float32))
    hyb_call = ((lambda: searcher.search_hybrid_batch(qb, sq_list, 10, 0.1)) if world > 1
                else (lambda: engine.search_hybrid_batch(qb, sq_list, 10, 0.1, raw=True)))
    hyb_batched, hyb_dt, recall_hybrid, hyb_sparse = None, float("nan"), None, None
    try:
        for _ in range(2):
            hyb_batched = hyb_call()
        barrier()
        st0 = engine.stats()
        t1 = time.perf_counter()
        for _ in range(batch_reps):
            hyb_batched = hyb_call()
        barrier()
        hyb_dt = time.perf_counter() - t1
        st1 = engine.stats()
        n_grouped = st1.get("sparse_grouped", 0) - st0.get("sparse_grouped", 0)
        # (the candidate count of a call is booked when the next one starts: the timed calls' counts are those of
        # batch_reps calls, shifted by one)
        hyb_sparse = {"queries_through_the_grouped_scan": n_grouped,
                      "queries_redone_on_the_per_query_kernels": st1.get("sparse_group_redo", 0) - st0.get("sparse_group_redo", 0),
                      "candidate_keys_ranked_per_query": round((st1.get("sparse_group_candidates", 0) - st0.get("sparse_group_candidates", 0))
                                                               / max(n_grouped, 1), 1)}
        # recall@10 against the single-query hybrid path (itself held to the oracle bit for bit by the tests)
        hits, n_ref = 0, min(100, args.queries)
        for i in range(n_ref):
            got_rows = hyb_batched[i][0] if world > 1 else hyb_batched[0][i, : hyb_batched[3][i]]
            hits += len(set(np.asarray(search(20 + i)[0]).tolist()) & set(np.asarray(got_rows).tolist()))
        recall_hybrid = hits / (10.0 * n_ref)
    except Exception as exc:
        print(f"[bench] hybrid batched-query section failed on rank {rank}: {exc!r}", file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([hyb_dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hyb_dt = float(t.item())
    qps_hybrid = batch_reps * args.queries / hyb_dt if hyb_dt == hyb_dt else None
    # the same query when it arrives as TEXT: WordPiece ids of a short question (12 tokens) are encoded by
    # the engine first (embed_query's path, embedding.py:76-86), then searched — single GPU only
    enc_lat = full_lat = None
    if world == 1:
        q_ids = np.concatenate([[101], np.random.default_rng(9).integers(1000, 29000, size=10), [102]]).astype(np.int32)
        q_off = np.asarray([0, q_ids.size], np.int32)
        e_l, f_l = np.empty(200), np.empty(200)
        for i in range(220):
            t1 = time.perf_counter()
            vec = enc.encode(engine, q_ids, q_off)[0]
            t2 = time.perf_counter()
            engine.search_hybrid(vec, q_terms[i % args.queries, : q_nnz[i % args.queries]], ones[: q_nnz[i % args.queries]], 10, 0.1)
            t3 = time.perf_counter()
            if i >= 20:
                e_l[i - 20], f_l[i - 20] = t2 - t1, t3 - t1
        enc_lat, full_lat = float(np.percentile(e_l, 50) * 1e3), float(np.percentile(f_l, 50) * 1e3)

    # ---- recall@10 of the dense scan against a plain torch f32 matmul over the same vectors ------
    recall = None
    if rank == 0:
        hits = 0
        nq = len(dense_only)
        xs = torch.cat(corpus_chunks)  # rows 0 .. corpus-1 of this rank's engine
        for i in range(nq):
            ref = torch.topk(xs @ qs[20 + i], 10).indices.cpu().numpy()
            got = dense_only[i][0][0]
            got = got[got < args.corpus]  # rows indexed by the timed steps are not in `xs`
            hits += len(set(ref.tolist()) & set(got.tolist()))
        recall = hits / (10.0 * nq)
        recall_batched = None
        if world == 1 and batched is not None:
            hits = 0
            for i in range(nq):
                ref = torch.topk(xs @ qs[20 + i], 10).indices.cpu().numpy()
                got = np.asarray(batched[0][i, : batched[2][i]])
                got = got[got < args.corpus]
                hits += len(set(ref.tolist()) & set(got.tolist()))
            recall_batched = hits / (10.0 * nq)
        del xs
    del corpus_chunks
    torch.cuda.empty_cache()

    # ---- the same dense search on an anisotropic corpus of the same size ---------------------------------------
    aniso = None
    if world == 1 and args.aniso_rows > 0:
        # (isotropic reference for the same kind of query: dense top-10, query vector given)
        iso_lat = np.empty(200)
        for i in range(200):
            t1 = time.perf_counter()
            engine.search_dense(qs_host[20 + i: 21 + i], 10)
            iso_lat[i] = time.perf_counter() - t1
        aniso = anisotropic_section(args, torch, dev, dim)
        aniso["isotropic_p50_dense_top10_ms"] = round(float(np.percentile(iso_lat, 50) * 1e3), 4)

    # ---- the reference's unmodified caller sequences on the native classes ---------
    dropin = None
    if world == 1 and args.dropin_files > 0:
        dropin = dropin_section(args, engine, state, np.random.default_rng(17))
    del state

    # ---- the other model widths of BASELINE.json, short runs (outside `value`) ------------------------------------
    other = None
    if world == 1 and args.other_rows > 0:
        other = []
        for m in OTHER_MODELS:
            try:
                other.append(other_width_section(args, torch, dev, m, args.other_rows))
            except Exception as exc:
                print(f"[bench] {m['name']} section failed: {exc!r}", file=sys.stderr, flush=True)

    if rank == 0:
        gemm_tf = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # ceiling for ALGORITHMIC FLOP/s: the f32 MFMA peak, the dense f16 peak, or a third of it (three passes)
        gemm_peak = {"f32": PEAK_F32_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS,
                     "f16x3": round(PEAK_F16_MFMA_TFLOPS / 3.0, 1)}[args.precision]
        scan_gbps = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        out = {
            "metric": "chunks indexed/sec + p50 top-10 query latency @1M-chunk corpus, 1/8 MI355X",
            "value": round(chunks_per_s, 1),
            "unit": "chunks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "f16x3": "f16x3 (operands as hi+lo f16 = 22 significant bits, three f16-MFMA passes, f32 "
                               "accumulate; measured |1-cos| ~5e-8 vs the f64 oracle)",
                      "f16": "f16 (f16 MFMA operands, f32 accumulate; softmax, LayerNorm statistics and pooling in f32; the "
                             "residual stream between layers is stored as f16 (LayerNorm is applied inside the GEMM epilogues from "
                             "f32 row sums); measured |1-cos| = 1.5e-6 vs the f64 oracle on THIS code path at this scale — 2300 "
                             "sequences / 272k tokens, 12 layers, sampled sequences incl. tile and forward-chunk edges "
                             "(tests/test_encoder_gpu.py::test_bench_scale_f16_path_against_the_f64_oracle; 4.1e-6 at 24 layers), "
                             "north_star tolerance 1e-4)"
                      }[args.precision],
            "data": "synthetic (seeded token ids / unit vectors; random-init N(0,0.02) weights of the named shape)",
            "config": {
                "workload": "BASELINE configs[2]: 1M-chunk corpus, bge-base-en-v1.5 shape (L12 H768 CLS) dense + "
                            "BM25 sparse, min-max hybrid top-10 (prefetch 30), 1 MI355X per shard",
                "chunks_per_step_per_gpu": args.batch,
                "tokens_per_chunk_mean": round(tokens / (args.steps * args.batch), 1),
                "corpus_rows_per_gpu": int(n_rows),
                "queries": args.queries,
                "fusion": "minmax (reference vector_store.py:659-689)",
                "parallelism": f"shard{world}",
            },
            "p50_query_ms": round(p50, 4),
            "p99_query_ms": round(p99, 4),
            "query_kind": "single-stream hybrid top-10 (dense top-30 + sparse top-30 + fusion), query vector given",
            "p50_query_encode_ms": None if enc_lat is None else round(enc_lat, 4),
            "p50_query_from_tokens_ms": None if full_lat is None else round(full_lat, 4),
            "query_from_tokens_kind": "the same search with the 12-token query embedded by the engine first (vr_encode + vr_search_hybrid)",
            "recall_at_10_dense_vs_torch_matmul": recall,
            "qps_batched_1k": None if qps_batched is None else round(qps_batched, 1),
            "batched_kind": f"{args.queries} dense top-10 queries per call (vr_search_dense, host arrays in and out), "
                            f"{batch_reps} calls timed; answers bit-identical to the single-query path (tests/test_search_gpu.py, "
                            "tests/test_fullsize_gpu.py)",
            "ms_per_batched_call": None if qps_batched is None else round(batch_dt / batch_reps * 1e3, 3),
            "qps_hybrid_batched_1k": None if qps_hybrid is None else round(qps_hybrid, 1),
            "ms_per_hybrid_batched_call": None if qps_hybrid is None else round(hyb_dt / batch_reps * 1e3, 3),
            "hybrid_batched_kind": f"{args.queries} hybrid top-10 queries per call (vr_search_hybrid_batch: one batched dense search, one "
                                   "batched sparse search over the inverted index beside it — the grouped scan of csrc/invert.hip: "
                                   "pairs of queries share a block per segment (four blocks per CU), thresholds from a sample of the segments — "
                                   "min-max fusion of every query on the host threads; host arrays in and out); answers "
                                   "bit-identical to the single-query path (tests/test_batch_hybrid_gpu.py, tests/test_fullsize_gpu.py)",
            "hybrid_batched_sparse_leg": hyb_sparse,
            "recall_at_10_hybrid_batched_vs_single_query": recall_hybrid,
            "other_model_widths": other,
            "recall_at_10_batched_vs_torch_matmul": recall_batched,
            "anisotropic_corpus": aniso,
            "dropin_index_chunks_per_s": None if dropin is None else round(dropin[0], 1),
            "dropin_sync_index_chunks_per_s": None if dropin is None else round(dropin[5], 1),
            "dropin_sync_kind": None if dropin is None else
                f"the same per-file sequence in the DEFAULT mode ({dropin[6]} files): embeddings stay token ids until store_chunks, which makes "
                "ONE fused engine call per file itself and raises where the reference's upsert raises (indexing.py:558-590 contract)",
            "dropin_kind": None if dropin is None else
                f"{dropin[2]} synthetic documents ({dropin[1]} chunks) from raw text through the reference's per-file sequence "
                "count_by_file -> delete_by_file -> chunk_text -> embed_texts -> sparse embed_texts -> zip -> store_chunks "
                "(indexing.py:239,284,513-563) on the drop-in "
                "classes, one thread, until every row is searchable (opt-in write-behind, VOITTA_DEFERRED_INDEXING=1: "
                "voitta_rag_amd/deferred.py; flush() + failed_file_paths() before the caller commits)",
            "p50_query_from_text_ms": None if dropin is None else round(dropin[3], 4),
            "p99_query_from_text_ms": None if dropin is None else round(dropin[4], 4),
            "query_from_text_kind": "embed_query(text) -> sparse embed_query(text) -> VectorStoreService.search(limit=10, hybrid) "
                                    "as mcp_server.py:469-485 calls them, StoredChunk objects out; corpus as above",
            # f32: algorithmic FLOP against the f32-MFMA peak. f16x3: every algorithmic multiply-add is
            # three f16 MFMA multiply-adds, so the ceiling for ALGORITHMIC FLOP/s is the dense f16 peak / 3.
            "roofline": {
                "kernel": {"f32": "vr::gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                           "f16x3": "vr::gemm_f16x3_256_kernel<EPI, 3> (v_mfma_f32_16x16x32_f16, 3 passes per "
                                    "product; 256x256 tiles)",
                           "f16": "vr::gemm_f16_pp_kernel<EPI> (v_mfma_f32_16x16x32_f16, one pass; 256x256x64 tiles, "
                                  "8 waves in two phase-shifted groups)"}[args.precision],
                "bound": "mfma",
                "achieved": round(gemm_tf, 2),
                "peak": gemm_peak,
                "unit": "TFLOP/s",
                "frac": round(gemm_tf / gemm_peak, 4),
                "executed_mfma_TFLOPs": round(gemm_tf * (3 if args.precision == "f16x3" else 1), 1),
                "traffic": pmc_traffic({"f32": "gemm_f32_kernel", "f16x3": "gemm_f16x3_256_kernel",
                                        "f16": "gemm_f16_pp_kernel"}[args.precision]),
                "algorithmic_flop_per_launch": round(gemm_flop / max(gemm_n, 1)),
                "launches": gemm_n,
                "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                "share_of_step_time": round(gemm_ms * 1e-3 / dt, 4),
            },
            "roofline_search": {
                "kernel": ("vr::prefilter_scan_kernel (v_mfma_f32_16x16x32_f16 streaming scan of the f16 shadow "
                           if os.environ.get("VR_PREFILTER") == "f16" or MODEL["hidden"] % 64 else
                           "vr::prefilter_scan8_kernel (v_mfma_i32_16x16x64_i8 streaming scan of the int8 shadow ") +
                          "corpus, stage 1 of the exact two-stage search; timed with the sparse leg running "
                          "beside it on the second stream)",
                "bound": "hbm",
                "achieved": round(scan_gbps, 1),
                "peak": PEAK_HBM_GBPS,
                "unit": "GB/s",
                "frac": round(scan_gbps / PEAK_HBM_GBPS, 4),
                "traffic": pmc_traffic("prefilter_scan_kernel") or pmc_traffic("prefilter_scan8_kernel"),
                "two_stage": engine.stats(),
                "algorithmic_bytes_per_launch": round(scan_bytes / max(scan_n, 1)),
                "launches": scan_n,
                "avg_launch_ms": round(scan_ms / max(scan_n, 1), 4),
                "sparse_scan_avg_ms": round(sp_ms / max(sp_n, 1), 4),
                # null when the queries ran on the inverted index (csrc/invert.hip): it reads its terms' postings
                # only, a number the host does not know; VR_SPARSE_INVERTED=0 brings back the all-ids scan
                "sparse_scan_GBps": round(sp_bytes / (sp_ms * 1e-3) / 1e9, 1) if sp_ms > 0 and sp_bytes > 0 else None,
                "sparse_scan_kind": "inverted index (postings of the query's terms)" if sp_bytes == 0 and sp_n > 0
                else "forward SELL-64 scan",
            },
            "roofline_batched_search": {
                "kernel": "vr::batch_scan_kernel<1> (v_mfma_i32_16x16x64_i8: int8 shadow corpus x int8 query parts -> per-slab lower "
                          "bounds for the thresholds and an f16 upper bound per (16-row tile, query)) + batch_flag_kernel + "
                          "batch_pairs_kernel (the few (tile, query) pairs that reach a threshold, on v_dot4_i32_i8); timed "
                          "together; exact f32 re-score of the candidates follows",
                "bound": "mfma",
                "achieved": round(bscan_ops / (bscan_ms * 1e-3) / 1e12, 2) if bscan_ms > 0 else None,
                "peak": 2 * PEAK_F16_MFMA_TFLOPS,
                "unit": "TOP/s (int8; algorithmic 2*N*D*Q per launch)",
                "frac": round(bscan_ops / (bscan_ms * 1e-3) / 1e12 / (2 * PEAK_F16_MFMA_TFLOPS), 4) if bscan_ms > 0 else None,
                "launches": bscan_n,
                "avg_launch_ms": round(bscan_ms / max(bscan_n, 1), 4),
                "share_of_call_time": round(bscan_ms * 1e-3 / batch_dt, 4) if qps_batched is not None else None,
                "traffic": pmc_traffic("batch_scan_kernel"),
            },
            # attention reads Q, K, V once (f16) and writes the context rows: 4 H halfs per token and launch — at the
            # ~118-token sequences of the indexing path it is bound by that traffic, not by its 1.1 TFLOP per batch
            "roofline_attention": {
                "kernel": "vr::attention_seq_kernel<64> (one block per (sequence, head): K/V rows staged once, "
                          "v_mfma_f32_16x16x32_f16 for QK^T and PV, V^T fragments from ds_read_b64_tr_b16)",
                "bound": "hbm",
                "achieved": round(attn_bytes / (attn_ms * 1e-3) / 1e9, 1) if attn_ms > 0 else None,
                "peak": PEAK_HBM_GBPS,
                "unit": "GB/s",
                "frac": round(attn_bytes / (attn_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if attn_ms > 0 else None,
                "algorithmic_bytes_per_launch": round(attn_bytes / max(attn_n, 1)),
                "launches": attn_n,
                "avg_launch_ms": round(attn_ms / max(attn_n, 1), 4),
                "TFLOPs": round(attn_flop / (attn_ms * 1e-3) / 1e12, 2) if attn_ms > 0 else None,
                "share_of_step_time": round(attn_ms * 1e-3 / dt, 4),
                "traffic": pmc_traffic("attention_seq_kernel"),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, np.random.default_rng(3))
            out["speedup_vs_cpu_index"] = round(chunks_per_s / out["cpu_baseline"]["value"], 1)
            out["speedup_vs_cpu_query"] = round(out["cpu_baseline"]["query_p50_ms"] / p50, 1)
        print(json.dumps(out), flush=True)
    engine.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
