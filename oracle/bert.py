"""TEST INFRASTRUCTURE — NumPy restatement of what ``SentenceTransformer.encode`` computes for the
BERT-family checkpoints the reference loads (reference call sites:
src/voitta/services/embedding.py:40,53,68-73,85; model choice src/voitta/config.py:33-34).

The arithmetic lives in sentence-transformers / transformers / torch (pyproject.toml:29-34,
un-vendored, lower-bounded only). Restated from their published behaviour [EXT], SURVEY.md a4:
  embeddings : LayerNorm(word[id] + position[p] + token_type[0]), eps = 1e-12
  per layer  : q,k,v = xW^T + b ; softmax(q k^T / sqrt(d_h)) v over the sequence's own tokens
               (padding never contributes: HF adds finfo.min to masked logits, exp underflows to 0)
               x = LayerNorm(ctx W_o^T + b_o + x)
               x = LayerNorm(GELU_erf(x W_1^T + b_1) W_2^T + b_2 + x)
  pooling    : mean over the sequence's tokens (sum / max(count, 1e-9)) or CLS (token 0)
  normalize  : x / max(||x||_2, 1e-12)
Every sequence is processed unpadded; sorting by length and the batch size of 32
(embedding.py:56,70) do not change results.

Pinned here against transformers.BertModel with seeded random weights
(tests/golden/make_bert_golden.py -> tests/golden/bert_*.npz). PARITY UNPINNED against the real
checkpoints: no weights exist offline and the reference's tests hold no embedding fixture.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class BertShape:
    layers: int
    hidden: int
    heads: int
    intermediate: int
    vocab: int = 30522
    max_pos: int = 512
    type_vocab: int = 2
    eps: float = 1e-12


# standard published shapes [EXT]
SHAPES = {
    "all-MiniLM-L6-v2": (BertShape(6, 384, 12, 1536), "mean"),
    "bge-base-en-v1.5": (BertShape(12, 768, 12, 3072), "cls"),
    "bge-large-en-v1.5": (BertShape(24, 1024, 16, 4096), "cls"),
    "e5-base-v2": (BertShape(12, 768, 12, 3072), "mean"),
}


def layer_keys(i: int) -> dict[str, str]:
    p = f"encoder.layer.{i}."
    return {
        "q_w": p + "attention.self.query.weight", "q_b": p + "attention.self.query.bias",
        "k_w": p + "attention.self.key.weight", "k_b": p + "attention.self.key.bias",
        "v_w": p + "attention.self.value.weight", "v_b": p + "attention.self.value.bias",
        "o_w": p + "attention.output.dense.weight", "o_b": p + "attention.output.dense.bias",
        "ln1_g": p + "attention.output.LayerNorm.weight", "ln1_b": p + "attention.output.LayerNorm.bias",
        "i_w": p + "intermediate.dense.weight", "i_b": p + "intermediate.dense.bias",
        "f_w": p + "output.dense.weight", "f_b": p + "output.dense.bias",
        "ln2_g": p + "output.LayerNorm.weight", "ln2_b": p + "output.LayerNorm.bias",
    }


EMB_KEYS = {
    "word": "embeddings.word_embeddings.weight",
    "pos": "embeddings.position_embeddings.weight",
    "type": "embeddings.token_type_embeddings.weight",
    "ln_g": "embeddings.LayerNorm.weight",
    "ln_b": "embeddings.LayerNorm.bias",
}


def random_weights(shape: BertShape, seed: int, std: float = 0.02) -> dict[str, np.ndarray]:
    """Seeded BERT-init-like weights (N(0, std); LayerNorm gains near 1) in HF state-dict naming.
    Biases and LayerNorm parameters are randomised too so that every term is exercised."""
    rng = np.random.default_rng(seed)
    H, I = shape.hidden, shape.intermediate
    w = {
        EMB_KEYS["word"]: rng.normal(0, std, (shape.vocab, H)),
        EMB_KEYS["pos"]: rng.normal(0, std, (shape.max_pos, H)),
        EMB_KEYS["type"]: rng.normal(0, std, (shape.type_vocab, H)),
        EMB_KEYS["ln_g"]: 1.0 + rng.normal(0, 0.1, H),
        EMB_KEYS["ln_b"]: rng.normal(0, 0.1, H),
    }
    for i in range(shape.layers):
        k = layer_keys(i)
        for n, shp in (("q", (H, H)), ("k", (H, H)), ("v", (H, H)), ("o", (H, H)), ("i", (I, H)), ("f", (H, I))):
            w[k[n + "_w"]] = rng.normal(0, std * 2.5, shp)
            w[k[n + "_b"]] = rng.normal(0, 0.05, shp[0])
        for n in ("ln1", "ln2"):
            w[k[n + "_g"]] = 1.0 + rng.normal(0, 0.1, H)
            w[k[n + "_b"]] = rng.normal(0, 0.1, H)
    return {k: v.astype(np.float32) for k, v in w.items()}


_erf = np.vectorize(math.erf, otypes=[np.float64])


def _gelu(x):
    return (0.5 * x * (1.0 + _erf(x.astype(np.float64) / math.sqrt(2.0)))).astype(x.dtype)


def _ln(x, g, b, eps):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def encode_one(w: dict, shape: BertShape, ids, dtype=np.float64) -> np.ndarray:
    """last_hidden_state [S, H] of one unpadded sequence."""
    W = lambda k: w[k].astype(dtype)  # noqa: E731
    ids = np.asarray(ids, dtype=np.int64)
    S, H, nh = len(ids), shape.hidden, shape.heads
    dh = H // nh
    x = W(EMB_KEYS["word"])[ids] + W(EMB_KEYS["pos"])[:S] + W(EMB_KEYS["type"])[0]
    x = _ln(x, W(EMB_KEYS["ln_g"]), W(EMB_KEYS["ln_b"]), dtype(shape.eps))
    for i in range(shape.layers):
        k = layer_keys(i)
        q = (x @ W(k["q_w"]).T + W(k["q_b"])).reshape(S, nh, dh).transpose(1, 0, 2)
        kk = (x @ W(k["k_w"]).T + W(k["k_b"])).reshape(S, nh, dh).transpose(1, 0, 2)
        v = (x @ W(k["v_w"]).T + W(k["v_b"])).reshape(S, nh, dh).transpose(1, 0, 2)
        s = q @ kk.transpose(0, 2, 1) / dtype(math.sqrt(dh))
        s = s - s.max(axis=-1, keepdims=True)
        p = np.exp(s)
        p = p / p.sum(axis=-1, keepdims=True)
        ctx = (p @ v).transpose(1, 0, 2).reshape(S, H)
        x = _ln(ctx @ W(k["o_w"]).T + W(k["o_b"]) + x, W(k["ln1_g"]), W(k["ln1_b"]), dtype(shape.eps))
        h = _gelu(x @ W(k["i_w"]).T + W(k["i_b"]))
        x = _ln(h @ W(k["f_w"]).T + W(k["f_b"]) + x, W(k["ln2_g"]), W(k["ln2_b"]), dtype(shape.eps))
    return x


def sentence_embeddings(w: dict, shape: BertShape, seqs, pooling: str = "mean", normalize: bool = True,
                        dtype=np.float64) -> np.ndarray:
    out = np.zeros((len(seqs), shape.hidden), dtype=dtype)
    for i, ids in enumerate(seqs):
        h = encode_one(w, shape, ids, dtype)
        if pooling == "cls":
            e = h[0]
        else:
            e = h.sum(axis=0) / max(float(len(ids)), 1e-9)
        if normalize:
            e = e / max(float(np.sqrt((e * e).sum())), 1e-12)
        out[i] = e
    return out
