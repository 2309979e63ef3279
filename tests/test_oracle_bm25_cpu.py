"""Pins oracle/bm25.py (restatement of fastembed's Qdrant/bm25) with the known answers of
tests/golden/bm25_kat.json, and oracle_core's IDF / sparse scoring with hand-derived values."""
import json
import math
import os

import numpy as np

from oracle import bm25 as obm
from oracle import core as ocore

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bm25_kat.json")))


def test_porter2_published_vocabulary():
    bad = {w: (obm.stem(w), s) for w, s in KAT["stem"].items() if obm.stem(w) != s}
    assert not bad


def test_murmur3_reference_values():
    for s, want in KAT["murmur3_abs"].items():
        assert obm.token_id(s) == want
    for s, want in KAT["murmur3_signed"].items():
        h = obm.murmur3_32(s.encode())
        assert (h - (1 << 32) if h & 0x80000000 else h) == want


def test_tf_hand_derived():
    for row in KAT["tf"]:
        stems = ["a"] * row["count"] + [f"w{i}" for i in range(row["doc_len"] - row["count"])]
        m = obm.term_frequency(stems)
        assert abs(m[obm.token_id("a")] - row["tf"]) < 1e-12
        idx, val = obm.tf_from_hashed([obm.token_id(s) for s in stems])
        assert abs(val[idx.index(obm.token_id("a"))] - row["tf"]) < 1e-12


def test_idf_hand_derived():
    for row in KAT["idf"]:
        assert abs(ocore.idf(row["n"], row["df"]) - row["idf"]) < 1e-6
        assert abs(ocore.idf(row["n"], row["df"]) - math.log(1 + (row["n"] - row["df"] + 0.5) / (row["df"] + 0.5))) < 1e-6


def test_pipeline_semantics():
    # punctuation and case vanish, stop-words drop, "_" alone is a punctuation token, long tokens drop
    text = "The QUICK, brown fox's jumps!! over _ the lazy-dog " + "x" * 41 + " naïve café 123"
    st = obm.stems(text)
    assert st == ["quick", "brown", "fox", "jump", "lazi", "dog", "naïv", "café", "123"]
    idx, val = obm.embed([text])[0]
    assert len(idx) == len(set(idx)) == 9 and all(v > 0 for v in val)
    qi, qv = obm.query_embed("the quick quick foxes")
    assert qv == [1.0, 1.0] and sorted(qi) == sorted({obm.token_id("quick"), obm.token_id("fox")})
    assert obm.embed([""]) == [([], [])] and obm.query_embed("the of and") == ([], [])


def test_sparse_scoring_hand_case():
    # 3 docs; term 7 in docs 0,1 ; term 9 in doc 1 only. N=3.
    rows = [([7], [1.5]), ([7, 9], [1.0, 2.0]), ([11], [1.0])]
    sc = ocore.sparse_scores(rows, [9, 7], [1.0, 1.0])
    idf7 = np.float32(math.log(1 + (3 - 2 + 0.5) / (2 + 0.5)))
    idf9 = np.float32(math.log(1 + (3 - 1 + 0.5) / (1 + 0.5)))
    assert np.isclose(sc[0], idf7 * np.float32(1.5), rtol=1e-6)
    assert np.isclose(sc[1], idf7 * np.float32(1.0) + idf9 * np.float32(2.0), rtol=1e-6)
    assert sc[2] == -np.inf
