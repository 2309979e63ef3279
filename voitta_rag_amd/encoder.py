"""Host side of the dense encoder: hands BERT-family weights (held by PyTorch or NumPy) to
``vr_encoder_load`` in the order the C-ABI fixes, and drives ``vr_encode``.

Replaces the ``SentenceTransformer(model_name, device)`` object the reference lazily creates
(src/voitta/services/embedding.py:23-42) — model architecture facts come from the checkpoint's
``config.json`` / ``modules.json``; PyTorch is used only to read and hold the weights."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import VR_MEM_DEVICE, VR_MEM_HOST, VR_POOL_CLS, VR_POOL_MEAN, check

EMB_SUFFIXES = [
    "embeddings.word_embeddings.weight",
    "embeddings.position_embeddings.weight",
    "embeddings.token_type_embeddings.weight",
    "embeddings.LayerNorm.weight",
    "embeddings.LayerNorm.bias",
]
LAYER_SUFFIXES = [
    "attention.self.query.weight", "attention.self.query.bias",
    "attention.self.key.weight", "attention.self.key.bias",
    "attention.self.value.weight", "attention.self.value.bias",
    "attention.output.dense.weight", "attention.output.dense.bias",
    "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
    "intermediate.dense.weight", "intermediate.dense.bias",
    "output.dense.weight", "output.dense.bias",
    "output.LayerNorm.weight", "output.LayerNorm.bias",
]


@dataclass
class BertDesc:
    layers: int
    hidden: int
    heads: int
    intermediate: int
    vocab: int = 30522
    max_pos: int = 512
    type_vocab: int = 2
    pooling: str = "mean"  # "mean" | "cls"
    normalize: bool = True
    eps: float = 1e-12
    precision: str = "f32"  # "f32" (exact f32 MFMA) | "f16x3" (split-precision f16 MFMA, f32-class accuracy)

    def to_c(self) -> _lib.VrBertDesc:
        d = _lib.VrBertDesc()
        d.struct_size = C.sizeof(_lib.VrBertDesc)
        d.layers, d.hidden, d.heads, d.intermediate = self.layers, self.hidden, self.heads, self.intermediate
        d.vocab, d.max_pos, d.type_vocab = self.vocab, self.max_pos, self.type_vocab
        d.pooling = VR_POOL_CLS if self.pooling == "cls" else VR_POOL_MEAN
        d.normalize = int(self.normalize)
        d.eps = self.eps
        codes = {"f32": _lib.VR_PRECISION_F32, "f16x3": _lib.VR_PRECISION_F16X3, "f16": _lib.VR_PRECISION_F16}
        if self.precision not in codes:
            raise ValueError(f"unknown encoder precision {self.precision!r}")
        d.precision = codes[self.precision]
        return d


def tensor_names(layers: int) -> list[str]:
    names = list(EMB_SUFFIXES)
    for i in range(layers):
        names += [f"encoder.layer.{i}.{s}" for s in LAYER_SUFFIXES]
    return names


def expected_shape(desc: BertDesc, name: str) -> tuple:
    """Shape of a state-dict entry as the description implies it ([out, in] layout, as HF stores it)."""
    H, inter = desc.hidden, desc.intermediate
    if name.endswith("word_embeddings.weight"): return (desc.vocab, H)
    if name.endswith("position_embeddings.weight"): return (desc.max_pos, H)
    if name.endswith("token_type_embeddings.weight"): return (desc.type_vocab, H)
    if name.endswith("intermediate.dense.weight"): return (inter, H)
    if name.endswith("intermediate.dense.bias"): return (inter,)
    if name.endswith("output.dense.weight") and "attention" not in name: return (H, inter)
    if name.endswith(".weight") and "LayerNorm" not in name: return (H, H)
    return (H,)


def _find(state: dict, suffix: str):
    if suffix in state:
        return state[suffix]
    hits = [k for k in state if k.endswith("." + suffix)]
    if len(hits) != 1:
        raise KeyError(f"weight '{suffix}' not found (or ambiguous) in state dict: {hits[:3]}")
    return state[hits[0]]


def load_encoder(engine, desc: BertDesc, state: dict) -> None:
    """state: HF BertModel state dict (any key prefix), values NumPy arrays or torch tensors.
    Tensors on the engine's GPU are passed as device pointers, everything else as host memory."""
    names = tensor_names(desc.layers)
    tensors = [_find(state, n) for n in names]
    # raw pointers cross the C-ABI next: a checkpoint whose config.json disagrees with its weights (padded or
    # resized vocabulary, another max_position_embeddings, ...) must fail HERE, not read past a buffer there
    for name, t in zip(names, tensors):
        want = expected_shape(desc, name)
        if tuple(t.shape) != want:
            raise ValueError(f"weight '{name}' has shape {tuple(t.shape)}, the model description implies {want}")
    on_device = all(hasattr(t, "is_cuda") and t.is_cuda for t in tensors)
    keep, ptrs = [], []
    for t in tensors:
        if on_device:
            import torch

            t = t.detach().to(torch.float32).contiguous()
            ptrs.append(t.data_ptr())
        else:
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            t = np.ascontiguousarray(t, dtype=np.float32)
            ptrs.append(t.ctypes.data)
        keep.append(t)
    if on_device:
        engine._follow(keep[0])
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    cdesc = desc.to_c()
    check(engine._lib.vr_encoder_load(engine.handle, C.byref(cdesc), arr, len(ptrs),
                                      VR_MEM_DEVICE if on_device else VR_MEM_HOST))
    engine.encoder_desc = desc


def encode(engine, ids, offsets, out=None):
    """ids/offsets: NumPy int32 (host) or torch int32 tensors on the GPU. Returns an (n, H) f32
    NumPy array, or fills/returns ``out`` when a device tensor is given."""
    desc = engine.encoder_desc
    dev_in = hasattr(ids, "is_cuda") and ids.is_cuda
    if dev_in:
        engine._follow(ids)
        n = int(offsets.shape[0]) - 1
        ip, op_ = C.c_void_p(ids.data_ptr()), C.c_void_p(offsets.data_ptr())
        mem = VR_MEM_DEVICE
    else:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = offsets.shape[0] - 1
        ip, op_ = C.c_void_p(ids.ctypes.data), C.c_void_p(offsets.ctypes.data)
        mem = VR_MEM_HOST
    if out is not None and hasattr(out, "is_cuda") and out.is_cuda:
        assert out.is_contiguous() and tuple(out.shape) == (n, desc.hidden)
        engine._follow(out)
        check(engine._lib.vr_encode(engine.handle, ip, op_, n, mem, C.c_void_p(out.data_ptr()), VR_MEM_DEVICE))
        return out
    res = np.empty((n, desc.hidden), np.float32)
    check(engine._lib.vr_encode(engine.handle, ip, op_, n, mem, C.c_void_p(res.ctypes.data), VR_MEM_HOST))
    return res
