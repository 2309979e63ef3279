"""HIP encoder (through vr_encoder_load / vr_encode) against (a) the golden embeddings produced by
transformers.BertModel + sentence-transformers pooling/normalise on torch-CPU f32 and (b) the
NumPy f64 restatement. Tolerance: BASELINE.json north_star asks for embedding cosine within 1e-4
of the reference CPU path; an all-f32 path should be ~1e-7, so the test pins 1e-5 and reports the
worst case."""
import glob
import os

import numpy as np
import pytest

from oracle import bert as obert

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bert_*.npz")))
COS_TOL = 1e-5  # |1 - cos(ours, reference)|; north_star allows 1e-4
ABS_TOL = 2e-5  # per-component, unit-length embeddings


def _case(path):
    from test_oracle_bert_cpu import load_case

    return load_case(path)


def _encode(shape, pooling, w, seqs, precision="f32"):
    from voitta_rag_amd import Engine
    from voitta_rag_amd import encoder as enc

    e = Engine(shape.hidden)
    desc = enc.BertDesc(shape.layers, shape.hidden, shape.heads, shape.intermediate, vocab=shape.vocab,
                        max_pos=shape.max_pos, type_vocab=shape.type_vocab, pooling=pooling, normalize=True,
                        eps=shape.eps, precision=precision)
    enc.load_encoder(e, desc, w)
    ids = np.concatenate(seqs).astype(np.int32)
    off = np.zeros(len(seqs) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in seqs])
    out = enc.encode(e, ids, off)
    e.close()
    return out


def _check(got, want, what, precision="f32"):
    cos = (got * want).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(want, axis=1)
    worst = float(np.max(np.abs(1.0 - cos)))
    print(f"{what}: worst |1-cos| = {worst:.3e}, worst abs diff = {np.max(np.abs(got - want)):.3e}")
    cos_tol, abs_tol = TOL[precision]
    assert worst < cos_tol
    assert np.max(np.abs(got - want)) < abs_tol


# f32 and f16x3 hold the same bar; plain f16 operands (11 significant bits) are
# held to the north_star tolerance itself, 1e-4 on the cosine, with a tenfold margin
PRECISIONS = ["f32", "f16x3", "f16"]
TOL = {"f32": (COS_TOL, ABS_TOL), "f16x3": (COS_TOL, ABS_TOL), "f16": (1e-5, 5e-4)}


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("path", [p for p in GOLDEN if "tiny" not in p], ids=lambda p: os.path.basename(p))
def test_encoder_matches_transformers_golden(gpu, path, precision):
    shape, pooling, seed, seqs, want = _case(path)
    w = obert.random_weights(shape, seed)
    got = _encode(shape, pooling, w, seqs, precision)
    _check(got, want, os.path.basename(path) + " vs transformers f32 " + precision, precision)
    _check(got, obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64), "vs numpy f64 " + precision, precision)


@pytest.mark.parametrize("name,layers,lens", [
    ("all-MiniLM-L6-v2", 6, [1, 2, 17, 64, 65, 128, 129, 200, 256, 31, 90, 90, 127]),
    ("bge-base-en-v1.5", 2, [512, 3, 130, 64, 333]),
    ("bge-large-en-v1.5", 1, [77, 512, 128]),
    ("e5-base-v2", 1, [100, 110, 120, 130]),
    ("bge-base-en-v1.5", 12, [118, 64]),     # full depth: the error the bench configuration really has
    ("bge-large-en-v1.5", 24, [96]),
])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_encoder_full_width_shapes(gpu, name, layers, lens, precision):
    """Real model widths (layers reduced so the CPU oracle stays in seconds), ragged lengths that
    cross the 64-key tile and 128-row GEMM tile edges."""
    base, pooling = obert.SHAPES[name]
    shape = obert.BertShape(layers, base.hidden, base.heads, base.intermediate, vocab=1000, max_pos=base.max_pos)
    w = obert.random_weights(shape, 99)
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
    got = _encode(shape, pooling, w, seqs, precision)
    want = obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64)
    _check(got, want, name + " " + precision, precision)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


def test_f16x3_survives_awkward_magnitudes(gpu):
    """Weights spanning many binades and activations far from O(1): the per-tensor power-of-two
    scaling and the (hi, lo) split must keep f32-class accuracy."""
    base, pooling = obert.SHAPES["all-MiniLM-L6-v2"]
    shape = obert.BertShape(2, base.hidden, base.heads, base.intermediate, vocab=300, max_pos=128)
    w = obert.random_weights(shape, 17)
    rng = np.random.default_rng(3)
    for k in list(w):
        if k.endswith("dense.weight") or k.endswith("query.weight") or k.endswith("key.weight") or k.endswith("value.weight"):
            w[k] = (w[k] * np.exp2(rng.integers(-9, 3, size=w[k].shape))).astype(np.float32)
    w["embeddings.word_embeddings.weight"] = (w["embeddings.word_embeddings.weight"] * 300).astype(np.float32)
    seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in (5, 64, 128, 33)]
    want = obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64)
    _check(_encode(shape, pooling, w, seqs, "f32"), want, "awkward f32")
    _check(_encode(shape, pooling, w, seqs, "f16x3"), want, "awkward f16x3", "f16x3")
    _check(_encode(shape, pooling, w, seqs, "f16"), want, "awkward f16", "f16")


def test_encode_is_batch_invariant_and_chunked(gpu):
    """> 262144 tokens forces several forward chunks; every sequence must come out identical to
    encoding it alone (packed layout: no cross-sequence leakage, no dependence on the GEMM tile a
    row lands in)."""
    base, pooling = obert.SHAPES["all-MiniLM-L6-v2"]
    shape = obert.BertShape(1, base.hidden, base.heads, base.intermediate, vocab=500, max_pos=256)
    w = obert.random_weights(shape, 3)
    rng = np.random.default_rng(8)
    lens = rng.integers(1, 257, size=2200).tolist()
    seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
    assert sum(lens) > 262144
    all_out = _encode(shape, pooling, w, seqs)
    for i in (0, 57, 199, 399, 2199):
        one = _encode(shape, pooling, w, [seqs[i]])
        assert np.array_equal(one[0], all_out[i])


def test_small_batches_replayed_as_graphs_are_bit_identical(gpu):
    """A small forward pass is captured into a hipGraph the second time its shape is seen
    (encoder.hip, encoder_encode). Eager, captured and replayed runs must agree bit for bit, for the
    same input and for other inputs of the same shape (token count, sequence count, longest length)."""
    from voitta_rag_amd import Engine
    from voitta_rag_amd import encoder as enc

    base, pooling = obert.SHAPES["bge-base-en-v1.5"]
    shape = obert.BertShape(2, base.hidden, base.heads, base.intermediate, vocab=400, max_pos=128)
    w = obert.random_weights(shape, 21)
    rng = np.random.default_rng(4)

    def batch(lens):
        seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
        off = np.zeros(len(seqs) + 1, np.int32)
        off[1:] = np.cumsum(lens)
        return np.concatenate(seqs), off

    def engine():
        e = Engine(shape.hidden)
        enc.load_encoder(e, enc.BertDesc(shape.layers, shape.hidden, shape.heads, shape.intermediate, vocab=shape.vocab,
                                         max_pos=shape.max_pos, pooling=pooling, precision="f16"), w)
        return e

    e = engine()
    a_ids, a_off = batch([12])
    b_ids, b_off = batch([12])            # same shape, other tokens
    c_ids, c_off = batch([7, 12, 3])
    d_ids, d_off = batch([12, 3, 7])      # same (T, n_seq, max_len), other offsets
    first = {k: enc.encode(e, *v) for k, v in (("a", (a_ids, a_off)), ("c", (c_ids, c_off)))}      # eager
    second = {k: enc.encode(e, *v) for k, v in (("a", (a_ids, a_off)), ("c", (c_ids, c_off)))}     # captured + launched
    third = {k: enc.encode(e, *v) for k, v in (("a", (a_ids, a_off)), ("c", (c_ids, c_off)))}      # replayed
    for k in first:
        assert np.array_equal(first[k], second[k]) and np.array_equal(first[k], third[k])
    got_b, got_d = enc.encode(e, b_ids, b_off), enc.encode(e, d_ids, d_off)                         # replays of a's / c's graph
    fresh = engine()
    assert np.array_equal(got_b, enc.encode(fresh, b_ids, b_off))                                   # eager in a fresh engine
    assert np.array_equal(got_d, enc.encode(fresh, d_ids, d_off))
    # a big batch reallocates the workspace (graphs are dropped), then the small shape works again
    big_ids, big_off = batch([100] * 40)
    enc.encode(e, big_ids, big_off)
    assert np.array_equal(enc.encode(e, a_ids, a_off), first["a"])
    assert np.array_equal(enc.encode(e, a_ids, a_off), first["a"])
    e.close()
    fresh.close()


@pytest.mark.parametrize("name,layers,lens", [
    ("bge-base-en-v1.5", 3, [12]),           # CLS pooling: last layer runs its [CLS]-only tail
    ("bge-base-en-v1.5", 12, [7]),
    ("e5-base-v2", 3, [16]),                 # mean pooling: the very last LayerNorm stays a launch of its own
    ("e5-base-v2", 2, [5, 4, 3, 2, 2]),      # five short sequences, 16 tokens
    ("all-MiniLM-L6-v2", 6, [9]),
    ("bge-large-en-v1.5", 3, [11]),
])
def test_single_query_path_folds_layernorm_into_the_projections(gpu, name, layers, lens):
    """<= 16 tokens in f16 mode: the LayerNorm launches are folded into the Q/K/V and FFN-up projections
    (gemm_f16_skinny_ln_kernel). Same tolerance as every other f16 path against the f64 oracle, and the
    embeddings agree with those of the same sequences encoded inside a larger batch (unfolded path) to 1e-6."""
    base, pooling = obert.SHAPES[name]
    shape = obert.BertShape(layers, base.hidden, base.heads, base.intermediate, vocab=1000, max_pos=base.max_pos)
    w = obert.random_weights(shape, 17)
    rng = np.random.default_rng(8)
    seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
    got = _encode(shape, pooling, w, seqs, "f16")
    want = obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64)
    _check(got, want, f"{name} folded LN, {sum(lens)} tokens", "f16")
    padded = seqs + [rng.integers(0, shape.vocab, size=40).astype(np.int32)]
    other = _encode(shape, pooling, w, padded, "f16")[: len(seqs)]
    cos = (got * other).sum(1)
    assert np.max(np.abs(1.0 - cos)) < 1e-6


@pytest.mark.parametrize("name,layers,lens", [
    ("bge-base-en-v1.5", 3, [120, 97, 64, 33, 128, 101]),   # CLS: last layer runs its [CLS]-only tail
    ("e5-base-v2", 3, [128, 90, 77, 110, 45]),              # mean pooling: the very last LayerNorm stays a kernel
    ("bge-large-en-v1.5", 2, [100, 128, 60, 75]),
    ("bge-base-en-v1.5", 12, [128, 128, 64]),               # full depth through the folded path
    ("all-MiniLM-L6-v2", 6, [128, 99, 64, 31, 77]),         # H = 384: a 256-column tile and a half one
])
def test_large_batches_fold_layernorm_into_the_gemms(gpu, name, layers, lens):
    """More than 256 tokens in f16 mode: no LayerNorm pass between the GEMMs — the producing epilogue stores
    f16 pre-LN rows and partial row sums, the consuming GEMM carries the LayerNorm gain in its weights and
    applies (mean, 1/sigma) in its epilogue (EPI_FOLD_*). Same bar against the f64 oracle as every f16 path."""
    base, pooling = obert.SHAPES[name]
    shape = obert.BertShape(layers, base.hidden, base.heads, base.intermediate, vocab=1000, max_pos=base.max_pos)
    w = obert.random_weights(shape, 23)
    rng = np.random.default_rng(9)
    seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
    assert sum(lens) > 256
    got = _encode(shape, pooling, w, seqs, "f16")
    want = obert.sentence_embeddings(w, shape, seqs, pooling, True, np.float64)
    _check(got, want, f"{name} L{layers} folded GEMM LN, {sum(lens)} tokens", "f16")


@pytest.mark.parametrize("name,layers,n_seq", [
    ("bge-base-en-v1.5", 12, 2300),    # the bench's own step and a little more: 271k tokens = one full forward chunk + a tail
    ("bge-large-en-v1.5", 24, 600),    # 70k tokens, 24 layers
    ("all-MiniLM-L6-v2", 6, 800),      # 94k tokens; H = 384 (one and a half column tiles), mean pooling
])
def test_bench_scale_f16_path_against_the_f64_oracle(gpu, name, layers, n_seq):
    """The code path bench.py times — the persistent 256x256 ping-pong GEMM walking many rounds of tiles per launch,
    LayerNorm folded into the GEMMs (fold_big), the f16 residual stream, per-sequence attention, 262144-token
    forward chunks — held to the f64 oracle AT ITS OWN SCALE. An embedding depends on its own sequence only, so a
    sample is enough: first and last sequences, the ones whose rows straddle a 256-row tile edge, the ones on both
    sides of the 262144-token chunk edge, and a spread in between (reference: embedding.py:56-74; north_star
    tolerance 1e-4 on the cosine, held here to 1e-5)."""
    base, pooling = obert.SHAPES[name]
    shape = obert.BertShape(layers, base.hidden, base.heads, base.intermediate, vocab=1000, max_pos=base.max_pos)
    w = obert.random_weights(shape, 31)
    rng = np.random.default_rng(12)
    lens = rng.integers(96, 141, size=n_seq)
    seqs = [rng.integers(0, shape.vocab, size=int(n)).astype(np.int32) for n in lens]
    cu = np.concatenate([[0], np.cumsum(lens)])
    total = int(cu[-1])
    assert total > 65536
    got = _encode(shape, pooling, w, seqs, "f16")
    assert got.shape == (n_seq, shape.hidden) and np.all(np.isfinite(got))
    pick = {0, 1, n_seq - 2, n_seq - 1}
    # sequences whose rows cross a 256-row GEMM tile edge (the first few, one in the middle, the last)
    crossing = [i for i in range(n_seq) if cu[i] // 256 != (cu[i + 1] - 1) // 256]
    pick.update(crossing[:3] + crossing[len(crossing) // 2: len(crossing) // 2 + 2] + crossing[-3:])
    if total > 262144:  # both sides of the forward-chunk edge (encoder.hip: kMaxChunkTokens)
        edge = int(np.searchsorted(cu, 262144, side="right")) - 1  # first sequence of the second chunk
        pick.update(range(max(edge - 2, 0), min(edge + 2, n_seq)))
    want_n = 28 if layers <= 12 else 14
    pick.update(int(i) for i in rng.choice(n_seq, size=max(want_n - len(pick), 0), replace=False))
    pick = sorted(pick)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    want = obert.sentence_embeddings(w64, shape, [seqs[i] for i in pick], pooling, True, np.float64)
    _check(got[pick], want, f"{name} L{layers} f16 at {total} tokens, {len(pick)} sampled sequences", "f16")
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)


def test_f16_large_batch_rows_do_not_depend_on_their_place_in_the_batch(gpu):
    """The f16 form of test_encode_is_batch_invariant_and_chunked: > 262144 tokens (several forward chunks) through
    the large-batch kernels. A sequence's embedding must be the same BITS wherever it sits — other tile, other
    round of the persistent walk, other forward chunk — because a GEMM row depends on its own row only (the K order
    is the same for every row of every tile). The short-batch kernels differ by f16-level rounding, so the batch is
    compared with a rotated copy of itself (and, to 1e-5 on the cosine, with each sampled sequence encoded alone)."""
    base, pooling = obert.SHAPES["all-MiniLM-L6-v2"]
    shape = obert.BertShape(2, base.hidden, base.heads, base.intermediate, vocab=500, max_pos=256)
    w = obert.random_weights(shape, 3)
    rng = np.random.default_rng(8)
    lens = rng.integers(1, 257, size=2200).tolist()
    seqs = [rng.integers(0, shape.vocab, size=n).astype(np.int32) for n in lens]
    assert sum(lens) > 262144
    all_out = _encode(shape, pooling, w, seqs, "f16")
    shift = 777
    rotated = _encode(shape, pooling, w, seqs[shift:] + seqs[:shift], "f16")
    assert np.array_equal(np.concatenate([rotated[-shift:], rotated[:-shift]]), all_out)
    for i in (0, 57, 199, 399, 2199):
        one = _encode(shape, pooling, w, [seqs[i]], "f16")
        assert abs(1.0 - float((one[0] * all_out[i]).sum())) < 1e-5
