"""f1 — ChunkingService (reference: src/voitta/services/chunking.py). PARITY UNPINNED: the reference
module cannot be imported here (python-dotenv is absent) and holds no fixtures; the hand-derived
known answers in tests/golden/chunking_kat.json check the restatement, and the native chunker behind
vr_chunk_texts is checked against the restatement on adversarial text. No GPU needed."""
import json
import os
import random

import pytest

from oracle import chunking as ochunk
from voitta_rag_amd.chunking import Chunk, ChunkingService, get_chunking_service

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chunking_kat.json"), encoding="utf-8"))["cases"]


def native(text, size, overlap, strategy):
    return [(c.text, c.index, c.start_char, c.end_char) for c in ChunkingService(size, overlap, strategy).chunk_text(text)]


@pytest.mark.parametrize("case", KAT, ids=[f"{c['strategy']}-{i}" for i, c in enumerate(KAT)])
def test_known_answers(case):
    want = [tuple(c) for c in case["chunks"]]
    args = (case["text"], case["chunk_size"], case["chunk_overlap"], case["strategy"])
    assert ochunk.chunk_text(*args) == want
    assert native(*args) == want


ALPHABET = ["a", "b", "Z", "é", "ß", "日", "本", "𝒳", "\U0001F600", " ", " ", " ", "\n", "\n\n", "\t", " ", " ",
            "　", "\x1c", "\x85", ". ", "? ", "! ", "; ", ", ", ".", "?", "!", ".\n", "  ", "é"]


def random_text(rng, n):
    return "".join(rng.choice(ALPHABET) for _ in range(n))


@pytest.mark.parametrize("strategy", ["recursive", "sentence", "fixed"])
def test_native_matches_restatement_on_adversarial_text(strategy):
    rng = random.Random(1234 + len(strategy))
    for trial in range(400):
        text = random_text(rng, rng.choice([0, 1, 3, 17, 60, 200, 900]))
        size = rng.choice([1, 2, 5, 16, 64, 300])
        overlap = rng.choice([1, 2, 7, 50, 400])
        try:
            want = ochunk.chunk_text(text, size, overlap, strategy)
        except ValueError:
            # chunk_overlap >= chunk_size and a text that falls through to the size windows: the reference
            # never terminates there (chunking.py:187); the native side fails the call
            with pytest.raises(RuntimeError):
                native(text, size, overlap, strategy)
            continue
        assert native(text, size, overlap, strategy) == want, (trial, repr(text), size, overlap)


def test_prose_document_all_strategies():
    rng = random.Random(7)
    words = ["index", "vector", "the", "a", "retrieval", "query,", "chunk.", "Sparse", "dense;", "score!", "why?", "fusion"]
    paras = []
    for _ in range(40):
        paras.append(" ".join(rng.choice(words) for _ in range(rng.randint(5, 120))))
    text = "\n\n".join(paras) + "\n"
    for strategy in ("recursive", "sentence", "fixed"):
        for size, overlap in ((512, 50), (128, 16), (40, 39)):
            got = native(text, size, overlap, strategy)
            assert got == ochunk.chunk_text(text, size, overlap, strategy)
            assert [c[1] for c in got] == list(range(len(got)))
            assert all(c[0] == c[0].strip() and c[0] for c in got)
    # defaults: CHUNK_SIZE 512, CHUNK_OVERLAP 50, recursive (config.py:39-41); every chunk fits unless overlap was carried
    svc = get_chunking_service()
    assert (svc.chunk_size, svc.chunk_overlap, svc.strategy) == (512, 50, "recursive")
    chunks = svc.chunk_text(text)
    assert isinstance(chunks[0], Chunk) and max(len(c.text) for c in chunks) <= 512 + 50
    assert [(c.text, c.index, c.start_char, c.end_char) for c in chunks] == ochunk.chunk_text(text)


def test_batch_form_and_falsy_arguments():
    svc = ChunkingService(60, 0, "")  # 0 and "" fall back to the settings, as `x or settings.x` does
    assert (svc.chunk_size, svc.chunk_overlap, svc.strategy) == (60, 50, "recursive")
    texts = ["one two three four five six seven eight nine ten", "", "   ", "tiny", "x" * 95, None]
    per_doc = svc.chunk_texts(texts)
    assert len(per_doc) == len(texts) and per_doc[1] == [] and per_doc[2] == [] and per_doc[5] == []
    for t, chunks in zip(texts, per_doc):
        assert [(c.text, c.index, c.start_char, c.end_char) for c in chunks] == ochunk.chunk_text(t or "", 60, 0, "")
    assert svc.chunk_texts([]) == []


def test_overlap_not_smaller_than_size_is_refused_for_windows():
    with pytest.raises(RuntimeError):
        ChunkingService(4, 4, "fixed").chunk_text("abcdefghij")
    with pytest.raises(ValueError):
        ochunk.chunk_text("abcdefghij", 4, 4, "fixed")
