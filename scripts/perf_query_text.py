"""Where a question-from-text spends its time (the MCP tool's three calls on the drop-in classes, mcp_server.py:469-485):
python scripts/perf_query_text.py [rows] — cProfile of 300 questions over a bge-base-shaped random encoder."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
from voitta_rag_amd import Engine  # noqa: E402
from voitta_rag_amd import encoder as enc  # noqa: E402

M = bench.MODEL
engine = Engine(M["hidden"], initial_rows=rows + 4096)
state = bench.random_state(torch, gen, dev)
enc.load_encoder(engine, enc.BertDesc(M["layers"], M["hidden"], M["heads"], M["intermediate"], vocab=M["vocab"], max_pos=M["max_pos"],
                                      pooling=M["pooling"], precision="f16"), state)
bench.populate(torch, gen, dev, engine, rows, M["hidden"])


class Args:
    dropin_files = 0
    precision = "f16"


import tempfile  # noqa: E402

from voitta_rag_amd import config, embedding, sparse_embedding, store_registry, vector_store  # noqa: E402
from voitta_rag_amd.wordpiece import WordPieceTokenizer  # noqa: E402

rng = np.random.default_rng(17)
vocab, words = bench.synthetic_vocab(rng, M["vocab"])
d = tempfile.mkdtemp(prefix="voitta-perf-")
open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8").write("\n".join(vocab) + "\n")
os.environ["EMBEDDING_DIMENSION"] = str(M["hidden"])
os.environ["EMBEDDING_MODEL"] = M["name"]
config.get_settings.cache_clear()
store_registry.set_engine(engine)
desc = enc.BertDesc(M["layers"], M["hidden"], M["heads"], M["intermediate"], vocab=M["vocab"], max_pos=M["max_pos"], pooling=M["pooling"],
                    normalize=True, precision="f16")
emb = embedding.EmbeddingService()
emb._model = embedding.NativeSentenceEncoder(engine, desc, state, WordPieceTokenizer.from_pretrained(d), M["max_pos"])
embedding._embedding_service = emb
sp = sparse_embedding.get_sparse_embedding_service()
vs = vector_store.VectorStoreService()
bench.mirror_engine_rows(vs, engine.count()[0])
warr = np.array(words)
questions = [" ".join(warr[rng.integers(0, len(warr), size=int(rng.integers(5, 12)))]) + "?" for _ in range(340)]


def ask(q):
    return vs.search(emb.embed_query(q), limit=10, sparse_query=sp.embed_query(q), sparse_weight=0.1)


for q in questions[:40]:
    ask(q)
lat = []
for q in questions[40:]:
    t = time.perf_counter()
    ask(q)
    lat.append(time.perf_counter() - t)
print(f"p50 from text {np.percentile(lat, 50) * 1e3:.4f} ms, p99 {np.percentile(lat, 99) * 1e3:.4f} ms")
# the engine call alone
tok = emb.model.tokenizer._h
lat2 = []
for q in questions[40:]:
    t = time.perf_counter()
    engine.query_text(tok, q, q, 512, 10, 0.1)
    lat2.append(time.perf_counter() - t)
print(f"p50 engine.query_text alone {np.percentile(lat2, 50) * 1e3:.4f} ms")
lat3 = []
ids, off = emb.model.tokenize([questions[50]])
for q in questions[40:]:
    t = time.perf_counter()
    enc.encode(engine, ids, off)
    lat3.append(time.perf_counter() - t)
print(f"p50 enc.encode ({len(ids)} tokens) alone {np.percentile(lat3, 50) * 1e3:.4f} ms")
pr = cProfile.Profile()
pr.enable()
for q in questions[40:]:
    ask(q)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
engine.close()
