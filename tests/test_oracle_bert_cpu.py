"""Pins oracle/bert.py (NumPy restatement of the sentence-transformers encode pipeline) against
the golden vectors produced by transformers.BertModel in the build container
(tests/golden/make_bert_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import bert as obert

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bert_*.npz")))


def load_case(path):
    z = np.load(path)
    L, H, nh, I, vocab, max_pos, tv = (int(v) for v in z["shape"])
    shape = obert.BertShape(L, H, nh, I, vocab=vocab, max_pos=max_pos, type_vocab=tv, eps=float(z["eps"]))
    lens = z["lens"].tolist()
    ids = z["ids"]
    seqs, p = [], 0
    for n in lens:
        seqs.append(ids[p:p + n])
        p += n
    return shape, str(z["pooling"]), int(z["seed"]), seqs, z["embeddings"]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_numpy_restatement_matches_transformers(path):
    shape, pooling, seed, seqs, want = load_case(path)
    w = obert.random_weights(shape, seed)
    got64 = obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64)
    got32 = obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float32)
    # cosine of each embedding against the torch f32 result: the north_star tolerance is 1e-4
    for got in (got64, got32):
        cos = (got * want).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(want, axis=1)
        assert np.all(np.abs(1.0 - cos) < 1e-6), cos
        assert np.max(np.abs(got - want)) < 2e-5
    assert np.allclose(np.linalg.norm(got64, axis=1), 1.0, atol=1e-12)


def test_golden_files_present():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_torch_cpu_baseline_matches_golden(path):
    """oracle/bert_torch.py is what bench.py times as cpu_baseline: it must compute the same thing."""
    from oracle import bert_torch

    shape, pooling, seed, seqs, want = load_case(path)
    got = bert_torch.TorchBert(obert.random_weights(shape, seed), shape, pooling).encode(seqs, batch_size=3)
    cos = (got * want).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(want, axis=1)
    assert np.all(np.abs(1.0 - cos) < 1e-6) and np.max(np.abs(got - want)) < 2e-5
