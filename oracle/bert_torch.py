"""TEST INFRASTRUCTURE — torch-CPU restatement of the sentence-transformers encode call, used as
bench.py's ``cpu_baseline`` ("port": the reference stack itself cannot be installed here).
Mirrors how SentenceTransformer.encode drives the model for EmbeddingService.embed_texts
(reference: src/voitta/services/embedding.py:68-73, batch_size=32 at :56) [EXT]:
sort by length (longest first), batches of 32, pad to the longest of the batch with an additive
attention mask, f32 on all host threads, mean/CLS pooling, L2 normalise, restore order.
Same arithmetic as oracle/bert.py (pinned against it in tests/test_oracle_bert_cpu.py)."""
from __future__ import annotations

import math

import numpy as np
import torch

from .bert import EMB_KEYS, BertShape, layer_keys


class TorchBert:
    def __init__(self, weights: dict, shape: BertShape, pooling: str = "mean", normalize: bool = True):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
        self.shape, self.pooling, self.normalize = shape, pooling, normalize

    @torch.no_grad()
    def _forward(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        w, sh = self.w, self.shape
        B, S = ids.shape
        nh, dh = sh.heads, sh.hidden // sh.heads
        F = torch.nn.functional
        x = w[EMB_KEYS["word"]][ids] + w[EMB_KEYS["pos"]][:S][None] + w[EMB_KEYS["type"]][0][None, None]
        x = F.layer_norm(x, (sh.hidden,), w[EMB_KEYS["ln_g"]], w[EMB_KEYS["ln_b"]], sh.eps)
        bias = (1.0 - mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
        for i in range(sh.layers):
            k = layer_keys(i)
            q = F.linear(x, w[k["q_w"]], w[k["q_b"]]).view(B, S, nh, dh).transpose(1, 2)
            kk = F.linear(x, w[k["k_w"]], w[k["k_b"]]).view(B, S, nh, dh).transpose(1, 2)
            v = F.linear(x, w[k["v_w"]], w[k["v_b"]]).view(B, S, nh, dh).transpose(1, 2)
            p = torch.softmax(q @ kk.transpose(-1, -2) / math.sqrt(dh) + bias, dim=-1)
            ctx = (p @ v).transpose(1, 2).reshape(B, S, sh.hidden)
            x = F.layer_norm(F.linear(ctx, w[k["o_w"]], w[k["o_b"]]) + x, (sh.hidden,), w[k["ln1_g"]], w[k["ln1_b"]], sh.eps)
            h = F.gelu(F.linear(x, w[k["i_w"]], w[k["i_b"]]))
            x = F.layer_norm(F.linear(h, w[k["f_w"]], w[k["f_b"]]) + x, (sh.hidden,), w[k["ln2_g"]], w[k["ln2_b"]], sh.eps)
        if self.pooling == "cls":
            e = x[:, 0]
        else:
            mf = mask.unsqueeze(-1).to(x.dtype)
            e = (x * mf).sum(1) / torch.clamp(mf.sum(1), min=1e-9)
        if self.normalize:
            e = F.normalize(e, p=2, dim=1)
        return e

    def encode(self, seqs, batch_size: int = 32) -> np.ndarray:
        order = np.argsort([-len(s) for s in seqs], kind="stable")
        out = np.zeros((len(seqs), self.shape.hidden), np.float32)
        for a in range(0, len(seqs), batch_size):
            sel = order[a:a + batch_size]
            S = max(len(seqs[i]) for i in sel)
            ids = torch.zeros((len(sel), S), dtype=torch.long)
            mask = torch.zeros((len(sel), S), dtype=torch.long)
            for r, i in enumerate(sel):
                ids[r, : len(seqs[i])] = torch.from_numpy(np.asarray(seqs[i], dtype=np.int64))
                mask[r, : len(seqs[i])] = 1
            out[sel] = self._forward(ids, mask).numpy()
        return out
