"""The C++ host text pipeline of libvoitta_engine.so (vr_bm25_tokenize / vr_porter2_stem) against
the Python restatement of fastembed's Bm25 (oracle/bm25.py) and the published known answers.
Host-only entry points: no GPU needed."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import bm25 as obm
from voitta_rag_amd import bm25 as vbm

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bm25_kat.json")))


def test_porter2_known_answers():
    bad = {w: (vbm.stem(w), s) for w, s in KAT["stem"].items() if vbm.stem(w) != s}
    assert not bad


def test_murmur_through_tokenizer():
    for word, want in KAT["murmur3_abs"].items():
        if not word or word in obm.STOPWORDS:
            continue
        off, ids = vbm.hashed_stems([word])
        assert ids.tolist() == [obm.token_id(obm.stem(word))]
    off, ids = vbm.hashed_stems(["quick"])
    assert ids.tolist() == [KAT["murmur3_abs"]["quick"]]


TEXTS = [
    "", " ", "The QUICK, brown fox's jumps!! over _ the lazy-dog " + "x" * 41 + " naïve café 123",
    "İstanbul'da ÇAY içtik; ΣΊΣΥΦΟΣ ΟΔΥΣΣΕΥΣ straße STRASSE", "snake_case __dunder__ a_b _ __ x_",
    "日本語のテキスト と English mixed 文字列", "tabs\tand\nnewlines\r\nand\x0bvertical\x1cfs",
    "don't won't it's they're should've", "running runs ran easily fairly generously communication",
    "１２３ full-width ＡＢＣ ² ½ Ⅻ", "emoji 😀 test ✓ done", "a" * 40 + " " + "b" * 41,
    "é combining vs é precomposed", "ǅ titlecase ß ſ long-s K kelvin",
]


@pytest.mark.parametrize("text", TEXTS)
def test_pipeline_matches_python_restatement(text):
    off, ids = vbm.hashed_stems([text])
    assert ids.tolist() == obm.hashed_stems(text), obm.stems(text)


def test_batch_offsets():
    off, ids = vbm.hashed_stems(TEXTS)
    want = [obm.hashed_stems(t) for t in TEXTS]
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(w) for w in want])]).tolist()
    assert ids.tolist() == [t for w in want for t in w]


@settings(max_examples=300, deadline=None)
@given(st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=80))
def test_pipeline_fuzz(text):
    if "Σ" in text:  # capital sigma: str.lower()'s context rule is approximated (documented)
        return
    off, ids = vbm.hashed_stems([text])
    assert ids.tolist() == obm.hashed_stems(text)


@settings(max_examples=500, deadline=None)
@given(st.text(alphabet="abcdefghijklmnopqrstuvwxyz'", min_size=1, max_size=14))
def test_stemmer_fuzz_ascii(word):
    assert vbm.stem(word) == obm.stem(word)
