"""Generates tests/golden/bert_*.npz: seeded weights run through transformers.BertModel (torch
CPU, f32, eager attention) with the padded-batch + attention-mask call sentence-transformers makes,
followed by its Pooling / Normalize modules restated in torch. Run in the build container:
    python tests/golden/make_bert_golden.py
Records the library versions in the fixture. The weights themselves are NOT stored: they are
regenerated from the seed by oracle.bert.random_weights (same NumPy generator)."""
import os
import sys

import numpy as np
import torch
import transformers

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bert as obert  # noqa: E402

CASES = {
    # name: (shape, pooling, seed, sequence lengths)
    "tiny_mean": (obert.BertShape(2, 64, 4, 128, vocab=200, max_pos=64), "mean", 11, [1, 2, 5, 17, 33, 64, 16, 3]),
    "tiny_cls": (obert.BertShape(2, 64, 2, 256, vocab=200, max_pos=96), "cls", 12, [7, 96, 40, 1, 65]),
    "minilm_l2": (obert.BertShape(2, 384, 12, 1536, vocab=500, max_pos=256), "mean", 13, [12, 130, 77, 256, 31]),
    "base_l1": (obert.BertShape(1, 768, 12, 3072, vocab=300, max_pos=128), "cls", 14, [128, 19, 64]),
}


def hf_model(shape, weights):
    cfg = transformers.BertConfig(
        vocab_size=shape.vocab, hidden_size=shape.hidden, num_hidden_layers=shape.layers,
        num_attention_heads=shape.heads, intermediate_size=shape.intermediate,
        max_position_embeddings=shape.max_pos, type_vocab_size=shape.type_vocab,
        layer_norm_eps=shape.eps, hidden_act="gelu", hidden_dropout_prob=0.0,
        attention_probs_dropout_prob=0.0)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    m = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    sd = m.state_dict()
    for k, v in weights.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
    missing = [k for k in sd if k not in weights and "position_ids" not in k and "token_type_ids" not in k]
    assert not missing, missing
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=False)
    return m


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (shape, pooling, seed, lens) in CASES.items():
        w = obert.random_weights(shape, seed)
        rng = np.random.default_rng(seed + 1000)
        seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
        m = hf_model(shape, w)
        B, S = len(seqs), max(lens)
        ids = torch.zeros((B, S), dtype=torch.long)
        mask = torch.zeros((B, S), dtype=torch.long)
        for i, s in enumerate(seqs):
            ids[i, : len(s)] = torch.from_numpy(s.astype(np.int64))
            mask[i, : len(s)] = 1
        with torch.no_grad():
            h = m(input_ids=ids, attention_mask=mask).last_hidden_state
            if pooling == "cls":
                e = h[:, 0]
            else:  # sentence_transformers.models.Pooling, mean mode
                mf = mask.unsqueeze(-1).to(h.dtype)
                e = (h * mf).sum(1) / torch.clamp(mf.sum(1), min=1e-9)
            e = torch.nn.functional.normalize(e, p=2, dim=1)  # models.Normalize
        np.savez_compressed(
            os.path.join(out_dir, f"bert_{name}.npz"),
            shape=np.array([shape.layers, shape.hidden, shape.heads, shape.intermediate, shape.vocab,
                            shape.max_pos, shape.type_vocab]),
            eps=np.array(shape.eps), pooling=np.array(pooling), seed=np.array(seed),
            lens=np.array(lens), ids=np.concatenate(seqs), embeddings=e.numpy().astype(np.float32),
            versions=np.array(f"transformers {transformers.__version__}; torch {torch.__version__}; numpy {np.__version__}"))
        print(name, e.shape, "ok")


if __name__ == "__main__":
    main()
