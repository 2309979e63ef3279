"""vr_bm25_tf and the fused vr_index_batch (encode -> BM25 tf -> store) against the oracles.
Reference flow restated: IndexingService._index_file_standard, src/voitta/services/indexing.py:527-560
(embed_texts, sparse embed_texts, store_chunks), then VectorStoreService.search (:560-697)."""
import numpy as np
import pytest

from oracle import bert as obert
from oracle import bm25 as obm
from oracle import core as ocore
from oracle import fusion as ofus

pytestmark = pytest.mark.gpu

WORDS = ("vector database index retrieval query embedding sparse dense hybrid fusion ranking chunk document "
         "folder search engine kernel memory bandwidth wavefront matrix tile running jumped happily relational "
         "the of and to in is it that was for on are as with they be at one have this from").split()


def _texts(rng, n, lo=3, hi=90):
    out = []
    for _ in range(n):
        m = int(rng.integers(lo, hi + 1))
        out.append(" ".join(rng.choice(WORDS, size=m)) + ("." if m % 2 else "!"))
    return out


def test_bm25_tf_bit_exact(gpu):
    from voitta_rag_amd import Engine

    rng = np.random.default_rng(1)
    texts = _texts(rng, 300) + ["", "the of and", "solo", " ".join(["repeat"] * 700), " ".join(WORDS * 40)]
    streams = [obm.hashed_stems(t) for t in texts]
    assert max(len(s) for s in streams) > 1024  # exercises the beyond-LDS path
    off = np.zeros(len(streams) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in streams])
    ids = np.array([t for s in streams for t in s], np.int32)
    e = Engine(64)
    got = e.bm25_tf(off, ids)
    for d, s in enumerate(streams):
        wi, wv = obm.tf_from_hashed(s)
        assert got[d][0].tolist() == wi, d
        assert got[d][1].tolist() == wv, d  # f64, bit for bit
        # and the same set of (id, value) pairs fastembed would return (order aside)
        m = obm.term_frequency(obm.stems(texts[d]))
        assert dict(zip(got[d][0].tolist(), got[d][1].tolist())) == m
    # non-default parameters
    got2 = e.bm25_tf(off[:11], ids[: off[10]], k=0.9, b=0.4, avg_len=100.0)
    for d in range(10):
        assert got2[d][1].tolist() == obm.tf_from_hashed(streams[d], 0.9, 0.4, 100.0)[1]
    e.close()


def test_index_batch_matches_stepwise_oracle(gpu):
    """encode + tf + store fused on the GPU == oracle encode -> oracle tf -> oracle search."""
    from voitta_rag_amd import Engine
    from voitta_rag_amd import encoder as enc

    rng = np.random.default_rng(2)
    shape = obert.BertShape(2, 128, 4, 256, vocab=400, max_pos=128)
    w = obert.random_weights(shape, 21)
    n = 230
    texts = _texts(rng, n, 2, 60)
    seqs = [rng.integers(0, shape.vocab, size=int(rng.integers(2, 100))).astype(np.int32) for _ in range(n)]
    streams = [obm.hashed_stems(t) for t in texts]
    e = Engine(shape.hidden)
    enc.load_encoder(e, enc.BertDesc(shape.layers, shape.hidden, shape.heads, shape.intermediate, vocab=shape.vocab,
                                     max_pos=shape.max_pos, pooling="mean"), w)
    cuts = [0, 70, 71, 200, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        wp_off = np.zeros(b - a + 1, np.int32)
        wp_off[1:] = np.cumsum([len(s) for s in seqs[a:b]])
        bm_off = np.zeros(b - a + 1, np.int64)
        bm_off[1:] = np.cumsum([len(s) for s in streams[a:b]])
        first = e.index_batch(np.concatenate(seqs[a:b]), wp_off,
                              np.array([t for s in streams[a:b] for t in s], np.int32), bm_off)
        assert first == a
    assert e.count() == (n, n)

    # dense side: stored rows == encode() output run through the cosine preprocessing
    emb_gpu = enc.encode(e, np.concatenate(seqs), np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int32))
    stored = e.get_dense(np.arange(n))
    assert np.array_equal(stored, ocore.cosine_preprocess(emb_gpu))
    emb_ref = obert.sentence_embeddings(w, shape, seqs, "mean", True, np.float64)
    cos = (stored * emb_ref).sum(1) / np.linalg.norm(stored, axis=1)
    assert np.max(np.abs(1 - cos)) < 1e-5

    # sparse side: rows as the oracle would have produced them (f32 as stored by Qdrant)
    sp = []
    for s in streams:
        idx, val = obm.tf_from_hashed(s)
        sp.append((np.array(idx, np.int32), np.array(val, np.float64).astype(np.float32)))
    for qtext in ("vector database retrieval", "running kernels happily", "chunk", "zzz unknown"):
        qi, qv = obm.query_embed(qtext)
        if not qi:
            continue
        want = ocore.sparse_scores(sp, qi, qv)
        wr, ws = ocore.topk(want, 30)
        gr, gs = e.search_sparse(qi, qv, 30)
        assert np.array_equal(gr, wr) and np.array_equal(gs, ws)
        q = emb_gpu[5]
        dsc = ocore.dense_scores(ocore.cosine_preprocess(q[None]), stored)[0]
        dr, ds = ocore.topk(dsc, 30)
        fused = ofus.hybrid_fuse(list(zip(dr.tolist(), ds.tolist())), list(zip(wr.tolist(), ws.tolist())), 10, 0.1)
        rows, scores, _ = e.search_hybrid(q, qi, qv, 10, 0.1)
        assert rows.tolist() == [r for r, _, _ in fused] and scores.tolist() == [s for _, s, _ in fused]
    e.close()
