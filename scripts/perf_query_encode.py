"""Latency of encoding ONE short query (embed_query's path): seeded bge-base-shaped weights."""
import sys, time
import numpy as np, torch
from oracle import bert as obert
from voitta_rag_amd import Engine
from voitta_rag_amd import encoder as enc
name = sys.argv[1] if len(sys.argv) > 1 else "bge-base-en-v1.5"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
shape, pooling = obert.SHAPES[name]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
H, I = shape.hidden, shape.intermediate
state = {}
for n in enc.tensor_names(shape.layers):
    if n.endswith("word_embeddings.weight"): shp = (shape.vocab, H)
    elif n.endswith("position_embeddings.weight"): shp = (shape.max_pos, H)
    elif n.endswith("token_type_embeddings.weight"): shp = (shape.type_vocab, H)
    elif n.endswith("intermediate.dense.weight"): shp = (I, H)
    elif n.endswith("intermediate.dense.bias"): shp = (I,)
    elif n.endswith("output.dense.weight") and "attention" not in n: shp = (H, I)
    elif n.endswith(".weight") and "LayerNorm" not in n: shp = (H, H)
    else: shp = (H,)
    t = torch.randn(shp, device=dev, generator=g) * 0.02
    if "LayerNorm.weight" in n: t = t + 1.0
    state[n] = t
e = Engine(H)
enc.load_encoder(e, enc.BertDesc(shape.layers, H, shape.heads, I, pooling=pooling, precision=prec), state)
for ntok in (12, 32, 128):
    ids = np.random.default_rng(1).integers(0, shape.vocab, size=ntok).astype(np.int32)
    off = np.array([0, ntok], np.int32)
    for _ in range(5): enc.encode(e, ids, off)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); enc.encode(e, ids, off); ts.append(time.perf_counter() - t0)
    print(f"[{prec}] {name}: 1 sequence of {ntok} tokens: p50 {np.percentile(ts, 50) * 1e3:.3f} ms")
