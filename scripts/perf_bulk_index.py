"""End-to-end indexing rate FROM RAW TEXT through the drop-in classes: parsed files -> ChunkingService ->
WordPiece + BM25 tokenisers -> vr_index_batch -> host payload table, with a bge-base-shaped synthetic
checkpoint (seeded random weights, synthetic 30k vocabulary). Compares
  (a) the reference's per-file sequence chunk_text -> embed_texts -> sparse embed_texts -> store_chunks
      (IndexingService._index_file_standard, indexing.py:513-563) on the native services, and
  (b) BulkIndexer (voitta_rag_amd/indexer.py): cross-file batches, host stages on a producer thread.
usage: python scripts/perf_bulk_index.py [n_files=600] [layers=12]"""
import json, os, pathlib, sys, tempfile, time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import bert as obert  # seeded weights only (bench/test infrastructure)

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 600
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(0)
letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
words = sorted({"".join(rng.choice(letters, size=int(rng.integers(1, 7)))) for _ in range(40000)})
vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789.,!?:;'-")
vocab += ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + words
vocab = list(dict.fromkeys(vocab))
shape = obert.BertShape(layers, 768, 12, 3072, vocab=len(vocab), max_pos=512)
d = pathlib.Path(tempfile.mkdtemp()) / "bge-base-shaped"
(d / "1_Pooling").mkdir(parents=True)
(d / "config.json").write_text(json.dumps({
    "model_type": "bert", "hidden_size": 768, "num_hidden_layers": layers, "num_attention_heads": 12,
    "intermediate_size": 3072, "vocab_size": len(vocab), "max_position_embeddings": 512, "type_vocab_size": 2,
    "layer_norm_eps": 1e-12, "hidden_act": "gelu"}))
(d / "modules.json").write_text(json.dumps([
    {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
    {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
    {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}]))
(d / "1_Pooling" / "config.json").write_text(json.dumps({
    "word_embedding_dimension": 768, "pooling_mode_cls_token": True, "pooling_mode_mean_tokens": False}))
(d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 512, "do_lower_case": True}))
(d / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
from safetensors.numpy import save_file
save_file({("bert." + k): v for k, v in obert.random_weights(shape, 3).items()}, str(d / "model.safetensors"))
os.environ["EMBEDDING_MODEL"] = str(d)
os.environ["EMBEDDING_DIMENSION"] = "768"

from voitta_rag_amd import config, embedding, sparse_embedding, store_registry, vector_store
from voitta_rag_amd.chunking import get_chunking_service
from voitta_rag_amd.indexer import BulkIndexer, ParsedFile
from voitta_rag_amd.vector_store import ChunkMetadata

warr = np.array(words)
def document(paragraphs):
    out = []
    for _ in range(paragraphs):
        sents = [" ".join(rng.choice(warr, size=int(rng.integers(5, 25)))).capitalize() + "." for _ in range(int(rng.integers(1, 5)))]
        out.append(" ".join(sents))
    return "\n\n".join(out)
files = [ParsedFile(document(int(rng.integers(3, 40))), f"dir{i % 7}/f{i}.md", f"dir{i % 7}", f"dir{i % 7}", f"f{i}.md",
                    source_modified_at=1_700_000_000 + i) for i in range(n_files)]
chars = sum(len(f.content) for f in files)

def fresh():
    config.get_settings.cache_clear(); store_registry.reset()
    embedding._embedding_service = None; sparse_embedding._sparse_embedding_service = None; vector_store._vector_store = None
    emb, sp, vs = embedding.get_embedding_service(), sparse_embedding.get_sparse_embedding_service(), vector_store.get_vector_store()
    emb.embed_texts(["warm up the encoder. " * 20] * 64); sp.embed_texts(["warm up"])
    return emb, sp, vs

# (b) bulk
emb, sp, vs = fresh()
BulkIndexer(batch_chunks=4096).index_files(files[:40])  # warm-up (graphs, workspaces, table growth)
emb, sp, vs = fresh()
t0 = time.perf_counter(); counts = BulkIndexer(batch_chunks=4096).index_files(files); vs.client.sync(); t1 = time.perf_counter()
n = sum(counts.values())
tok = emb.model.tokenize([c.text for c in get_chunking_service().chunk_text(files[0].content)])
print(f"{n_files} files, {chars / 1e6:.1f} M chars -> {n} chunks ({chars / n:.0f} chars, ~{tok[1][-1] / (len(tok[1]) - 1):.0f} tokens each), {layers} layers")
print(f"(b) BulkIndexer            : {t1 - t0:6.2f} s = {n / (t1 - t0):8.0f} chunks/s from raw text")

# (a) the reference's per-file sequence on the native services: write-behind (default), then literal
def per_file(sub):
    emb, sp, vs = fresh()
    chunker = get_chunking_service()
    t0 = time.perf_counter(); m = 0
    for f in sub:
        chunks = chunker.chunk_text(f.content)
        texts = [c.text for c in chunks]
        e, s = emb.embed_texts(texts), sp.embed_texts(texts)
        vs.store_chunks([(c.text, v, ChunkMetadata(f.file_path, f.folder_path, f.index_folder, f.file_name, c.index, len(chunks),
                                                   c.start_char, c.end_char, "t", source_modified_at=f.source_modified_at))
                         for c, v in zip(chunks, e)], sparse_vectors=s)
        m += len(chunks)
    vs.client.sync(); t1 = time.perf_counter()  # (.client waits for the flusher)
    return m, t1 - t0

per_file(files[:40])
m, dt = per_file(files)
print(f"(a) per-file, write-behind  : {dt:6.2f} s = {m / dt:8.0f} chunks/s ({len(files)} files)")
os.environ["VOITTA_DEFERRED_INDEXING"] = "0"
sub = files[: max(40, n_files // 4)]
m, dt = per_file(sub)
print(f"(a0) per-file, list-of-floats: {dt:6.2f} s = {m / dt:8.0f} chunks/s ({len(sub)} files)")
