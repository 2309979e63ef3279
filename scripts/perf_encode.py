"""Ad-hoc encoder throughput probe: seeded random weights of a named shape, synthetic token ids."""
import sys
import time

import numpy as np
import torch

from oracle import bert as obert  # only for shape table + seeded weights (test/bench infrastructure)
from voitta_rag_amd import Engine
from voitta_rag_amd import encoder as enc

name = sys.argv[1] if len(sys.argv) > 1 else "bge-base-en-v1.5"
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
seq_len = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
shape, pooling = obert.SHAPES[name]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
names = enc.tensor_names(shape.layers)
state = {}
H, I = shape.hidden, shape.intermediate
for n in names:
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
rng = np.random.default_rng(1)
lens = np.full(n_seq, seq_len) if seq_len > 0 else rng.integers(90, 131, size=n_seq)
ids = torch.from_numpy(rng.integers(0, shape.vocab, size=int(lens.sum())).astype(np.int32)).to(dev)
off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
out = torch.empty((n_seq, H), device=dev)
enc.encode(e, ids, off, out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    enc.encode(e, ids, off, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
T = int(lens.sum())
flop = T * shape.layers * (24 * H * H + 4 * seq_len * H)
print(f"[{prec}] {name}: {n_seq} seqs x {seq_len} tok: {dt*1e3:.2f} ms/batch, {n_seq/dt:.0f} chunks/s, {flop/dt/1e12:.1f} TFLOP/s (algorithmic)")
