/*
 * voitta_engine.h — C-ABI of libvoitta_engine.so, the MI355X (gfx950) in-process
 * indexing-and-retrieval engine that replaces, for voitta-rag's hot path only,
 *   - sentence-transformers' encode          (reference: src/voitta/services/embedding.py:40-86)
 *   - fastembed's Qdrant/bm25 sparse vectors  (reference: src/voitta/services/sparse_embedding.py:18-50,
 *                                              scripts/build_sparse_vectors.py:124,170)
 *   - the Qdrant client calls on the path     (reference: src/voitta/services/vector_store.py:233-317
 *                                              upsert, :560-697 search / hybrid fusion, :462-530 filters,
 *                                              :319-434 deletes)
 *
 * The reference has no FFI of its own (it is 100 % Python); these entry points are what a ctypes
 * binding inside the three service classes would call. Plain pointers and sizes only — no torch,
 * numpy or C++ types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; vr_last_error() returns a thread-local
 *     message (the Python side raises RuntimeError with it — reference callers catch broadly:
 *     src/voitta/services/indexing.py:531-533,561-563).
 *   - `mem` arguments say where the caller's buffers live: VR_MEM_HOST (pageable host memory,
 *     e.g. numpy) or VR_MEM_DEVICE (HBM of the engine's device, e.g. a torch-ROCm tensor).
 *     Output buffers follow the same `mem` unless stated otherwise.
 *   - row ids are dense int64 indices into the engine's HBM-resident tables, assigned in upsert
 *     order; the UUID<->row map and the text payload stay in the Python host.
 *   - thread-safety: one engine may be called from any number of threads. Searches run CONCURRENTLY, each on
 *     its own HIP stream with its own staging area ("lanes", VR_SEARCH_LANES of them, default 4; further
 *     searches wait for a free one), and hold a shared lock on the index for their duration. Mutations
 *     (vr_upsert, vr_index_batch, vr_delete_rows, vr_compact, vr_load) and vr_encode are serialised among
 *     themselves and take the exclusive lock only to PUBLISH: the append of a batch, the tombstones of a
 *     delete, the pointer swap of a compaction (which builds its result beside the live index). A search
 *     therefore sees the state before or after a mutation, never a mixture, and waits for no encode and
 *     no compaction (the reference's callers: watcher.py:149-171,
 *     indexing.py:281-288, api/routes/folders.py:137-143 beside MCP search threads).
 *   - there is NO CPU fallback: creating an engine without a usable gfx950 device fails.
 */
#ifndef VOITTA_ENGINE_H
#define VOITTA_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_ABI_VERSION 1

#define VR_MEM_HOST   0
#define VR_MEM_DEVICE 1

/* "field absent" marker for the source_created_at / source_modified_at columns
 * (reference stores the keys only when non-None: vector_store.py:279-282; a point lacking the
 * key fails a range `must`). */
#define VR_TS_ABSENT INT64_MIN

/* fusion modes for vr_search_hybrid */
#define VR_FUSION_MINMAX 0 /* reference behaviour: vector_store.py:659-689 */
#define VR_FUSION_RRF    1 /* north_star extra mode; no reference oracle */

typedef struct vr_engine vr_engine;

typedef struct vr_config {
  int32_t struct_size;   /* = sizeof(vr_config) */
  int32_t device;        /* HIP device ordinal */
  int32_t dim;           /* dense dimension D (multiple of 16, <= 1024): settings.embedding_dimension, embedding.py:20 */
  int32_t flags;         /* VR_ENGINE_* bits */
  int64_t initial_rows;  /* capacity hint; tables grow by doubling */
} vr_config;

/* vr_config.flags.
 * By default a single-query dense search over a store with dim % 32 == 0 runs in two stages:
 * a reduced-precision shadow copy of the corpus is scanned on the matrix cores — int8 with one
 * scale per row when dim % 64 == 0 (1/4 of the bytes, +25 % memory), f16 otherwise or when the
 * environment has VR_PREFILTER=f16 (1/2 of the bytes, +50 % memory) — a rigorous per-row error
 * bound (the norm of the row's quantisation residual, Cauchy-Schwarz) turns the approximate
 * scores into lower/upper bounds, and every row whose upper bound reaches the k-th best lower
 * bound is re-scored with the exact f32 chain. Results are bit-identical to the one-stage exact
 * scan (tests run both); the flag below turns the shadow copy and the first stage off. */
#define VR_ENGINE_NO_PREFILTER 1

/* Search-time predicate; restates _build_filter, vector_store.py:462-530. All ids are the
 * host's dictionary ids of folder_path / index_folder strings (exact string equality in the
 * reference => integer equality here). Arrays are host pointers and are tiny. */
typedef struct vr_filter {
  int32_t struct_size;
  int32_t n_must_folder_sets;       /* number of `must` any-of sets on folder_path (0..2):
                                       folder_filter (:476-482) and include_folders (:484-490) */
  const int32_t* must_folder_ids;   /* concatenated sets */
  const int32_t* must_folder_off;   /* n_must_folder_sets+1 offsets into must_folder_ids */
  const int32_t* not_folder_ids;    /* exclude_folders on folder_path (:492-499) */
  int32_t n_not_folder;
  int32_t n_not_index_folder;
  const int32_t* not_index_folder_ids; /* exclude_index_folders on index_folder (:501-508) */
  int32_t has_date_start;           /* gte (:514-515) */
  int32_t has_date_end;             /* lte (:516-517) */
  int64_t date_start;
  int64_t date_end;
  int32_t date_field;               /* 0 = source_modified_at (default), 1 = source_created_at (:511-512) */
  int32_t reserved0;
} vr_filter;

/* ---- lifecycle ---------------------------------------------------------------------------- */
int  vr_abi_version(void);
const char* vr_last_error(void);
int  vr_engine_create(const vr_config* cfg, vr_engine** out);
void vr_engine_destroy(vr_engine* e);
/* block until every kernel queued by previous calls has finished (for timing) */
int  vr_sync(vr_engine* e);
/* the engine's hipStream_t, as an opaque pointer (bench.py records HIP events on it) */
void* vr_stream(vr_engine* e);
/* queue all further work on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)
 * so that VR_MEM_DEVICE buffers produced on that stream need no extra synchronisation.
 * NULL rebinds the engine's own stream, which is a blocking stream (ordered against the legacy
 * null stream). Device buffers handed to the engine must be ready on the bound stream. */
int  vr_set_stream(vr_engine* e, void* stream);

/* ---- dense encode: replaces SentenceTransformer(...).encode (embedding.py:40,53,68-73,85) ----- */
#define VR_POOL_MEAN 0
#define VR_POOL_CLS  1

/* BERT-family encoder description; values come from the checkpoint's config.json /
 * sentence-transformers modules.json (SURVEY.md §8 model table). */
typedef struct vr_bert_desc {
  int32_t struct_size;
  int32_t layers;
  int32_t hidden;        /* H: multiple of 128, <= 1024 */
  int32_t heads;         /* H / heads must be 32 or 64 */
  int32_t intermediate;  /* multiple of 128 */
  int32_t vocab;
  int32_t max_pos;       /* max_position_embeddings; also the longest accepted sequence */
  int32_t type_vocab;
  int32_t pooling;       /* VR_POOL_MEAN | VR_POOL_CLS (sentence-transformers Pooling module) */
  int32_t normalize;     /* 1 = L2-normalise (sentence-transformers Normalize module) */
  float   eps;           /* layer_norm_eps */
  int32_t precision;     /* VR_PRECISION_F32 | VR_PRECISION_F16X3 | VR_PRECISION_F16 */
} vr_bert_desc;

/* arithmetic of the encoder's matrix products:
 *   F32    every product on the f32-input MFMA (exact f32 fma chains) — 157 TFLOP/s peak
 *   F16X3  operands carried as (hi, lo) f16 pairs (22 significant bits), three f16-MFMA passes per
 *          product into an f32 accumulator — f32-class accuracy (measured |1-cos| < 1e-6 against
 *          the f32 path) at 16/3 of the f32-MFMA rate. Activations beyond f16's range (65504) are
 *          clamped. Attention, LayerNorm, GELU, pooling stay f32 in both modes.
 *   F16    operands rounded to f16 (weights pre-scaled by a power of two per tensor), one f16-MFMA pass,
 *          f32 accumulate; everything outside the matrix products stays f32. Measured |1-cos| vs the
 *          f64 oracle: see tests/test_encoder_gpu.py (north_star tolerance: 1e-4). */
#define VR_PRECISION_F32   0
#define VR_PRECISION_F16X3 1
#define VR_PRECISION_F16   2

/* tensors: 5 + 16*layers f32 arrays in `mem`, HF BertModel state-dict order and [out,in] layout:
 *   word_embeddings, position_embeddings, token_type_embeddings, embeddings.LayerNorm.{weight,bias},
 *   then per layer: query.{weight,bias}, key.{weight,bias}, value.{weight,bias},
 *   attention.output.dense.{weight,bias}, attention.output.LayerNorm.{weight,bias},
 *   intermediate.dense.{weight,bias}, output.dense.{weight,bias}, output.LayerNorm.{weight,bias}.
 * The engine copies (and re-packs) them; the caller's tensors may be freed afterwards.
 * Replaces EmbeddingService.model's lazy SentenceTransformer(...) load (embedding.py:23-42). */
int vr_encoder_load(vr_engine* e, const vr_bert_desc* desc, const void* const* tensors,
                    int32_t n_tensors, int mem);

/* ids: concatenated WordPiece ids of n_seq sequences ([CLS] ... [SEP] already added, truncated
 * to max_pos by the tokenizer); offsets: n_seq+1 int32, offsets[0] = 0. Both in `mem`.
 * out: n_seq x H f32 sentence embeddings in `out_mem`. */
int vr_encode(vr_engine* e, const int32_t* ids, const int32_t* offsets, int32_t n_seq, int mem,
              float* out, int out_mem);

/* Host-only WordPiece tokenizer (no engine, no GPU): the tokenise step of SentenceTransformer.encode
 * (embedding.py:40,68-73 -> [EXT] HF tokenizers: BertNormalizer, BertPreTokenizer, WordPiece,
 * "[CLS] $A [SEP]", right truncation — SURVEY.md §8 a4 step 2), producing the ids / offsets vr_encode
 * and vr_index_batch take.
 *   vocab_tokens   n_vocab NUL-terminated UTF-8 strings, id = index (vocab.txt line order); must
 *                  contain [UNK], [CLS], [SEP]
 *   lowercase      do_lower_case; strip_accents: 0 / 1, or -1 = follow lowercase (the HF default)
 *   texts[i]       text_lens[i] bytes of UTF-8 (malformed bytes are dropped like U+FFFD)
 *   max_len        longest sequence including [CLS] and [SEP] (max_seq_length)
 *   out_offsets    n_texts + 1; out_ids: at most `capacity` ids written; *needed: total count
 * Returns -2 (offsets and *needed valid) when the buffer is too small. Thread-safe: the tokenizer
 * object is read-only after creation. */
typedef struct vr_wordpiece vr_wordpiece;
int vr_wordpiece_create(const char* const* vocab_tokens, int32_t n_vocab, int32_t lowercase, int32_t strip_accents,
                        int32_t handle_chinese_chars, int32_t clean_text, vr_wordpiece** out);
void vr_wordpiece_destroy(vr_wordpiece* t);
int vr_wordpiece_encode(const vr_wordpiece* t, const char* const* texts, const int64_t* text_lens, int64_t n_texts,
                        int32_t max_len, int64_t* out_offsets, int32_t* out_ids, int64_t capacity, int64_t* needed);

/* ---- BM25 document side: replaces SparseTextEmbedding("Qdrant/bm25").embed's token-count / TF
 * weighting (sparse_embedding.py:25,49; scripts/build_sparse_vectors.py:124,170) -- */
/* tok_off: n_docs+1 offsets, tok_ids: abs(murmur3) of each stemmed token in text order (`mem`).
 * Outputs (`mem`), padded layout: document d's distinct term ids, ascending, are
 * out_idx[tok_off[d] .. tok_off[d] + out_cnt[d]) with tf weights (f64, the Python floats the
 * reference passes on) at the same positions of out_val; out_idx/out_val hold tok_off[n_docs]
 * entries. tf = c*(k+1) / (c + k*(1 - b + b*doc_len/avg_len)); fastembed defaults k=1.2 b=0.75
 * avg_len=256. */
int vr_bm25_tf(vr_engine* e, const int64_t* tok_off, const int32_t* tok_ids, int64_t n_docs, int mem,
               double k, double b, double avg_len,
               int32_t* out_cnt, int32_t* out_idx, double* out_val);

/* Host-only text side of fastembed's Bm25 (no engine, no GPU): remove_non_alphanumeric,
 * SimpleTokenizer, stop-word / length filter, Snowball English stemmer, abs(murmur3_x86_32)
 * (SURVEY.md a6/a7 [EXT]). texts[i]: lens[i] bytes of UTF-8. out_off: n+1 offsets; out_ids: the
 * hashed stems of every text in order, at most `cap` written; *out_needed: total count. Returns -2
 * (offsets and *out_needed valid, the leading `cap` ids written) when the buffer is too small: call
 * again with one of *out_needed ids. The same stream serves documents
 * (-> vr_bm25_tf) and queries (Bm25.query_embed = the set of these ids, all values 1.0). */
int vr_bm25_tokenize(const char* const* texts, const int64_t* lens, int64_t n,
                     int64_t* out_off, int32_t* out_ids, int64_t cap, int64_t* out_needed);
/* Snowball English (Porter2) stem of one lower-case UTF-8 word; NUL-terminated into out. */
int vr_porter2_stem(const char* word, int64_t len, char* out, int64_t cap);

/* ---- chunking: replaces ChunkingService.chunk_text (src/voitta/services/chunking.py:33-241,
 * called from services/indexing.py:380,515), the step in front of the path. Host
 * only, one document per host thread. texts[i]: text_lens[i] bytes of UTF-8. chunk_size and
 * chunk_overlap count CHARACTERS (code points, Python's len()). strategy: VR_CHUNK_*; any other
 * value chunks recursively, as the reference does for an unknown name. The call fails when
 * chunk_overlap >= chunk_size AND some text falls through to the fixed-size windows: the
 * reference's window loop never advances there (chunking.py:187) and hangs.
 * The result object holds, until vr_chunks_free:
 *   doc_off   n_texts + 1  chunk index range of each text (an empty / all-blank text has none)
 *   span      2 per chunk  (start_char, end_char) code-point offsets exactly as the reference
 *                          reports them (for overlapped chunks it does not advance start_char)
 *   text_off  n_chunks + 1 byte offsets into `text`, the stripped chunk texts back to back (UTF-8)
 * Chunk.index is the position inside its text's range. */
#define VR_CHUNK_RECURSIVE 0
#define VR_CHUNK_SENTENCE 1
#define VR_CHUNK_FIXED 2
typedef struct vr_chunks vr_chunks;
int vr_chunk_texts(const char* const* texts, const int64_t* text_lens, int64_t n_texts, int32_t chunk_size,
                   int32_t chunk_overlap, int32_t strategy, vr_chunks** out);
int vr_chunks_view(const vr_chunks* c, int64_t* n_chunks, const int64_t** doc_off, const int64_t** span,
                   const int64_t** text_off, const char** text);
void vr_chunks_free(vr_chunks* c);

/* ---- fused indexing step: the three starred calls of IndexingService._index_file_standard
 * (src/voitta/services/indexing.py:527-530,560) — embed_texts, sparse embed_texts, store_chunks —
 * without leaving HBM: encode (vr_encode) -> BM25 tf (vr_bm25_tf) -> store (vr_upsert).
 * wp_ids / wp_off (int32, n+1): WordPiece ids per chunk; bm_ids / bm_off (int64 offsets, n+1):
 * hashed stems per chunk, or both NULL for a dense-only index. Token arrays are in `mem`; the
 * payload columns are host arrays as in vr_upsert. Needs a loaded encoder with hidden == dim. */
int vr_index_batch(vr_engine* e, int64_t n, int mem,
                   const int32_t* wp_ids, const int32_t* wp_off,
                   const int32_t* bm_ids, const int64_t* bm_off,
                   double k, double b, double avg_len,
                   const int32_t* folder_id, const int32_t* index_folder_id,
                   const int64_t* created, const int64_t* modified,
                   int64_t* out_first_row);

/* ---- index: replaces VectorStoreService.store_chunks' client.upsert (vector_store.py:291-313)
 * and the Qdrant-side cosine normalisation on insert (SURVEY.md a10 [EXT]). ------------------- */
/* dense   : n x D f32 row-major
 * sp_off  : n+1 CSR offsets, sp_idx/sp_val : BM25 (token id, tf weight); NULL sp_off => rows carry
 *           no sparse vector (vector_store.py:299-300)
 * folder_id / index_folder_id : per-row dictionary ids (NULL => 0)
 * created / modified : per-row epochs or VR_TS_ABSENT (NULL => all absent)
 * out_first_row : rows [first, first+n) were assigned */
int vr_upsert(vr_engine* e, int64_t n, int mem,
              const float* dense,
              const int64_t* sp_off, const int32_t* sp_idx, const float* sp_val,
              const int32_t* folder_id, const int32_t* index_folder_id,
              const int64_t* created, const int64_t* modified,
              int64_t* out_first_row);

/* tombstone rows (host pointer); replaces the filtered client.delete of vector_store.py:340-352,
 * :378-390,:419-431 once the host has resolved the filter to rows. Already-dead rows are ignored;
 * document frequencies and the sparse point count are decremented. */
int vr_delete_rows(vr_engine* e, const int64_t* rows, int64_t n);

/* counters of the two-stage dense search (see VR_ENGINE_NO_PREFILTER) */
#define VR_STAT_TWO_STAGE 0        /* single-query dense searches served by f16 scan + exact re-score */
#define VR_STAT_FALLBACK 1         /* ... of which exceeded the re-score budget and were redone one-stage */
#define VR_STAT_LAST_CANDIDATES 2  /* rows re-scored by the last two-stage search */
#define VR_STAT_BATCHED 3          /* queries served by the batched (integer GEMM) dense search */
#define VR_STAT_BATCH_FALLBACK 4   /* ... of which exceeded their candidate budget and were redone alone */
#define VR_STAT_BATCH_CANDIDATES 6 /* rows re-scored exactly by the batched search, summed over its queries */
#define VR_STAT_SPARSE_GROUPED 7    /* sparse queries served by the grouped batch scan (groups of queries share a block per segment) */
#define VR_STAT_SPARSE_GROUP_REDO 8 /* queries that scan gave up (a candidate region overflowed) and the per-query kernels redid */
#define VR_STAT_SPARSE_GROUP_CANDIDATES 9 /* candidate keys the grouped scan's selections ranked (counted when the NEXT batch starts) */
#define VR_STAT_GENERATION 5       /* bumped whenever row numbers change meaning (vr_compact, vr_load): a host
                                      table keyed by row is valid for the generation it was built against */
int vr_stats(vr_engine* e, int32_t which, int64_t* out);

/* n_rows = rows ever assigned, n_live = not tombstoned (get_collection_info, vector_store.py:699-710) */
int vr_count(vr_engine* e, int64_t* n_rows, int64_t* n_live);

/* read back stored (normalised) dense rows — scroll(with_vectors=True) of
 * scripts/build_sparse_vectors.py:140-146. out: n x D f32 (host). */
int vr_get_dense(vr_engine* e, const int64_t* rows, int64_t n, float* out);

/* document frequency of token ids and the sparse point count N (Qdrant Modifier.IDF statistics,
 * vector_store.py:95-99). ids/out_df are host pointers. */
int vr_sparse_stats(vr_engine* e, const int32_t* ids, int32_t n, int32_t* out_df, int64_t* out_n_points);

/* ---- search: replaces client.query_points (vector_store.py:612-617, :640-656) ------------- */
/* q : nq x D f32 (`mem`); results (host): rows[nq*k] (-1 padded), scores[nq*k], counts[nq].
 * Exact f32 brute force over live rows that pass `filter` (NULL = none), score = k-ordered f32
 * fma chain of q_hat . x_hat, ties broken by the lower row id.
 * More than 16 queries at once (BASELINE configs[4]: 1k batched queries) are served by an int8
 * matrix-core GEMM over the shadow corpus with certain error bounds, followed by an exact re-score of
 * the candidates (csrc/batch.hip): same results, bit for bit, at a small fraction of nq single searches. */
int vr_search_dense(vr_engine* e, const float* q, int32_t nq, int mem, int32_t k,
                    const vr_filter* filter,
                    int64_t* rows, float* scores, int32_t* counts);

/* The same search with the results left as packed ranking keys, nq x k uint64 in `keys_mem` memory (host or
 * device): key = (order-preserving bits of the f32 score << 32) | (0xFFFFFFFF - row), descending, 0 = no result.
 *   score bits: u = key >> 32;  f32 bits = (u & 0x80000000) ? u ^ 0x80000000 : ~u;   row = 0xFFFFFFFF - (key & 0xFFFFFFFF)
 * A sharded caller (voitta_rag_amd/sharded.py) hands the device array of a whole query batch straight to its
 * RCCL all_gather — the per-shard top-k merge of SURVEY.md §8e — without a host round trip per query. */
int vr_search_dense_keys(vr_engine* e, const float* q, int32_t nq, int mem, int32_t k, const vr_filter* filter,
                         uint64_t* keys, int keys_mem);

/* one sparse query (host pointers): score(d) = sum_t (q_t * idf(t)) * d_t over shared terms in
 * ascending token-id order, idf(t) = ln(1 + (N - df_t + 0.5)/(df_t + 0.5)); rows sharing no
 * term are not returned. */
int vr_search_sparse(vr_engine* e, const int32_t* q_idx, const float* q_val, int32_t nnz,
                     int32_t k, int32_t weights_given, const vr_filter* filter,
                     int64_t* rows, float* scores, int32_t* count);
/* weights_given != 0: q_val[t] already is q_t * idf(t) and the engine applies no IDF. Used when
 * the corpus is sharded over several engines (one per GPU): the caller all-reduces the document
 * frequencies of the query terms (vr_sparse_stats) and computes idf with vr_idf, so that every
 * shard scores with the collection-wide statistic. */
float vr_idf(int64_t n_points, int32_t df);

/* ---- measurement: HIP-event timing of the engine's own kernels on its stream ---------------- */
#define VR_PROF_GEMM 0         /* encoder GEMMs; work = FLOP (2*M*N*K) */
#define VR_PROF_ATTENTION 1    /* work = FLOP (4 * H * sum len^2) */
#define VR_PROF_DENSE_SCAN 2   /* work = bytes (N*D*4 + masks + scores) */
#define VR_PROF_SPARSE_SCAN 3  /* work = bytes (4 per stored id + 5 per row) */
#define VR_PROF_BATCH_SCAN 4   /* batched dense search, both integer-GEMM passes; work = operations (2*N*D*Q) */
/* enable != 0 clears earlier records and starts recording one event pair per launch */
int vr_profile(vr_engine* e, int enable);
int vr_profile_read(vr_engine* e, int kernel_class, double* total_ms, int64_t* launches,
                    double* total_work);

/* hybrid: restates VectorStoreService._hybrid_search (vector_store.py:621-697): dense and sparse
 * top-(3*limit), min-max normalisation, (1-w)*d + w*s over the id union, top-`limit`.
 * out_scores are the fused scores as f64 (Python floats in the reference, :680,:694).
 * out_from_dense[i] = 1 when the row was in the dense list (:682-685). */
int vr_search_hybrid(vr_engine* e, const float* q, int mem,
                     const int32_t* q_idx, const float* q_val, int32_t nnz,
                     int32_t limit, double sparse_weight, int32_t fusion,
                     const vr_filter* filter,
                     int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count);

/* ---- many queries per call (BASELINE configs[4]: 1k batched hybrid queries; the caller that would batch is the MCP
 * search tool under load, mcp_server.py:469-485). The sparse queries of a batch are one CSR: query i's terms are
 * q_idx / q_val [q_off[i], q_off[i + 1]) (host pointers). Results are, bit for bit, those of nq single calls. ---- */

/* nq sparse queries (vector_store.py:647-656, once per query): rows[nq*k] (-1 padded), scores[nq*k], counts[nq].
 * Queries of up to 32 distinct terms share ONE launch over the inverted index (csrc/invert.hip: grid = segment
 * share x query); longer ones, k > 64 and collections kept on the forward scan are served one by one. */
int vr_search_sparse_batch(vr_engine* e, const int64_t* q_off, const int32_t* q_idx, const float* q_val, int32_t nq,
                           int32_t k, int32_t weights_given, const vr_filter* filter,
                           int64_t* rows, float* scores, int32_t* counts);

/* nq hybrid queries (_hybrid_search, vector_store.py:621-697, once per query): the dense legs as ONE batched dense
 * search, the sparse legs as ONE batched sparse search beside it (second stream), the fusion of every query on the
 * host threads. q: nq x D f32 (`mem`); sq_off may be NULL (no query has sparse terms).
 * out_rows / out_scores / out_from_dense: nq x limit (entries beyond out_counts[i] are unspecified). */
int vr_search_hybrid_batch(vr_engine* e, const float* q, int32_t nq, int mem,
                           const int64_t* sq_off, const int32_t* sq_idx, const float* sq_val,
                           int32_t limit, double sparse_weight, int32_t fusion, const vr_filter* filter,
                           int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_counts);

/* Both legs of nq hybrid queries as packed ranking keys (see vr_search_dense_keys), for a sharded caller: keys is
 * [nq][2][k] uint64 in `keys_mem` memory — per query its dense list, then its sparse list. weights_given as in
 * vr_search_sparse. The caller all-gathers the array, merges with vr_merge_keys and fuses with vr_fuse_batch
 * (fusion runs on MERGED lists, never per shard: vector_store.py:659-689). */
int vr_search_hybrid_keys(vr_engine* e, const float* q, int32_t nq, int mem,
                          const int64_t* sq_off, const int32_t* sq_idx, const float* sq_val,
                          int32_t k, int32_t weights_given, const vr_filter* filter, uint64_t* keys, int keys_mem);

/* The per-shard top-k merge of SURVEY.md §8e on the engine's own kernel: parts is [n_parts][n_lists][k] keys (`mem`:
 * the tensor an RCCL all_gather filled, or a host array), part p holding shard p's lists. Every list is merged over
 * the parts — score descending, then lower local row, then lower part, i.e. ascending global id row * n_parts + p —
 * into out_ids (global ids, -1 padded), out_scores and out_counts, host arrays [n_lists][k] / [n_lists].
 * n_parts * k <= 4096. */
int vr_merge_keys(vr_engine* e, const uint64_t* parts, int32_t n_parts, int32_t n_lists, int32_t k, int mem,
                  int64_t* out_ids, float* out_scores, int32_t* out_counts);

/* The fusion arithmetic for nq pairs of lists at once, on the host threads (vector_store.py:659-697 per query; the
 * last step of a batched or sharded hybrid search). d_* / s_*: [nq][k] with counts [nq] (s_* may be NULL);
 * out_*: [nq][limit], out_counts [nq]. fusion: VR_FUSION_*; json_scores as in vr_fuse_minmax. */
int vr_fuse_batch(const int64_t* d_rows, const float* d_scores, const int32_t* d_counts,
                  const int64_t* s_rows, const float* s_scores, const int32_t* s_counts,
                  int32_t nq, int32_t k, int32_t limit, double sparse_weight, int32_t fusion, int32_t json_scores,
                  int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_counts);

/* ---- collection-wide document frequencies on a sharded corpus ("one all-reduce (sum) of df deltas
 * after each upsert/delete batch"; Qdrant's Modifier.IDF statistic is collection-wide, vector_store.py:95-99).
 * Every shard's table holds the statistic of ALL shards: after a shard stored (or before it deletes) rows, it exports
 * their term ids, the shards exchange them (all_gather) and each applies the OTHERS' ids. A query then needs no
 * exchange of statistics: one collective (the result merge) per hybrid query. -------------------------------------- */

/* Term ids of the listed rows' sparse vectors in a fixed-stride layout: out[i * stride + j], -1 where row i has no
 * j-th entry, is tombstoned or carries no sparse vector; out (`mem`, room for cap ids) may be NULL to ask for the
 * stride alone. *stride = the widest stored sparse row; *n_points = listed rows that are live and carry a sparse
 * vector. rows is a host array. */
int vr_sparse_row_ids(vr_engine* e, const int64_t* rows, int64_t n, int32_t* out, int64_t cap, int mem,
                      int32_t* stride, int64_t* n_points);
/* df[id] += sign for every id >= 0 of ids (`mem`), sparse point count += sign * n_points: the statistics of rows
 * that live on other shards. sign = +1 (stored there) or -1 (deleted there). */
int vr_df_apply(vr_engine* e, const int32_t* ids, int64_t n_ids, int mem, int64_t n_points, int32_t sign);

/* A question as TEXT, answered in one call: the three calls of the MCP search tool (mcp_server.py:469-485:
 * embedding_service.embed_query, sparse_service.embed_query, vector_store.search) without the trips through the host
 * language between them. dense_text: the query as the encoder sees it (with its "query: " prefix for e5 models,
 * embedding.py:82-83), tokenised by `tokenizer` ([CLS] .. [SEP], truncated to max_len); sparse_text: the raw query for
 * the BM25 side (Bm25.query_embed: the set of its hashed stems, every value 1.0), or NULL / length 0 for a dense-only
 * search. Branch selection as VectorStoreService.search (vector_store.py:560-619): hybrid (prefetch 3 x limit, fusion)
 * when a stem survives, else the dense top-`limit` with its cosine scores widened to f64 (the caller applies the
 * REST/JSON transport). *out_hybrid says which ran. Needs a loaded encoder with hidden == dim. */
int vr_query_text(vr_engine* e, const vr_wordpiece* tokenizer, const char* dense_text, int64_t dense_len,
                  const char* sparse_text, int64_t sparse_len, int32_t max_len, int32_t limit, double sparse_weight,
                  int32_t fusion, const vr_filter* filter,
                  int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count, int32_t* out_hybrid);

/* Persistence. The reference's index survives a restart in Qdrant's volume
 * (docker-compose.yml:8-9; VectorStoreService._ensure_collection re-attaches, vector_store.py:75-115).
 * vr_save writes everything the device owns — tiled dense corpus, payload columns, tombstones,
 * SELL sparse index, document-frequency table — to ONE file (written to path + ".tmp", then renamed
 * over `path`); vr_load restores it into an EMPTY engine of the same dimension and rebuilds the
 * search shadow. Row numbers, scores and rankings after a load are identical to those before the
 * save. The file is checksummed; a truncated or corrupt file is refused. Payload text, point ids
 * and the folder dictionaries belong to the host (voitta_rag_amd/vector_store.py saves them beside it). */
int vr_save(vr_engine* e, const char* path);

/* Reclaim tombstoned rows. The reference deletes points in place while it
 * serves (watcher deletes, re-index = delete + insert, orphan purge: services/indexing.py:281-288,
 * 696-721,886-901, watcher.py:149-171) and Qdrant's optimiser compacts segments in the background;
 * here vr_delete_rows leaves tombstones and this call rebuilds every table without them.
 * Surviving rows keep their relative order and are renumbered 0..n_live-1:
 *   new_row_of_old  host int64[rows before the call] (may be NULL): new row, or -1 for a dropped row
 *   n_rows_after    rows (= live rows) after the call
 * Scores and rankings are unchanged (document frequencies and the sparse point count never
 * included deleted rows). The compacted tables are built beside the live ones; searches keep running
 * and only wait for the final exchange of pointers. VR_STAT_GENERATION is bumped by that exchange. */
int vr_compact(vr_engine* e, int64_t* new_row_of_old, int64_t* n_rows_after);
int vr_load(vr_engine* e, const char* path);

/* The fusion arithmetic alone (host, no GPU): what vector_store.py:659-697 does to two result
 * lists. Exposed so the reference's own two-list code path can be checked in isolation.
 * json_scores != 0 reproduces the REST transport of the reference (f32 score -> shortest decimal
 * -> Python float) instead of exact f32->f64 widening. */
int vr_fuse_minmax(const int64_t* d_rows, const float* d_scores, int32_t nd,
                   const int64_t* s_rows, const float* s_scores, int32_t ns,
                   int32_t limit, double sparse_weight, int32_t json_scores,
                   int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count);

/* Reciprocal-rank fusion of the same two lists (host, no GPU): the arithmetic of vr_search_hybrid's
 * VR_FUSION_RRF mode in isolation. The reference does not fuse this way — its own note says why
 * (vector_store.py:638-639: Qdrant's prefetch + fusion is RRF-only, hence the weighted min-max path) —
 * north_star names it, so it is offered and checked against oracle/fusion.py::rrf_fuse:
 * score(id) = sum over the lists holding id of 1 / (position + 2), position counted from 0 [EXT: the
 * Qdrant server's RRF]; ties in the fused score go to the lower row id. */
int vr_fuse_rrf(const int64_t* d_rows, int32_t nd, const int64_t* s_rows, int32_t ns, int32_t limit,
                int64_t* out_rows, double* out_scores, int32_t* out_from_dense, int32_t* out_count);

#ifdef __cplusplus
}
#endif
#endif /* VOITTA_ENGINE_H */
