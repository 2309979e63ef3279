"""TEST INFRASTRUCTURE — Python restatement of the reference's ChunkingService
(src/voitta/services/chunking.py:19-246). Only tests/ may import this module;
the product chunker is csrc/chunking.cpp behind vr_chunk_texts.

Written with Python's own str primitives (len, strip, split, slicing, re.split, str.find) so that
code-point counting, the white-space set and the separator semantics are the interpreter's, not a
second guess at them; every function names the reference lines it follows. Chunks are plain tuples
``(text, index, start_char, end_char)``.

PARITY UNPINNED: the reference module cannot be imported here (voitta.config imports python-dotenv,
which this container does not have) and the reference's tests hold no chunking fixture, so no
output of the reference itself backs this restatement. tests/golden/chunking_kat.json holds
hand-derived cases that follow the reference source by hand, nothing more.
"""
from __future__ import annotations

import re

# chunking.py:52-62, most to least meaningful; "" means "cut by size"
SEPARATORS = ("\n\n", "\n", ". ", "? ", "! ", "; ", ", ", " ", "")
DEFAULTS = {"chunk_size": 512, "chunk_overlap": 50, "strategy": "recursive"}  # config.py:39-41


def _append(out: list, body: str, start: int, end: int) -> None:
    """The guarded append every branch shares (chunking.py:80-89,117-126,158-168,177-185)."""
    kept = body.strip()
    if kept:
        out.append((kept, len(out), start, end))


def _windows(body: str, origin: int, size: int, overlap: int, out: list) -> None:
    """chunking.py:170-191 (_split_by_size)."""
    if overlap >= size:
        raise ValueError("chunk_overlap >= chunk_size: the reference never terminates here (chunking.py:187)")
    at = 0
    while at < len(body):
        stop = min(at + size, len(body))
        _append(out, body[at:stop], origin + at, origin + stop)
        at += size - overlap


def _descend(body: str, seps: tuple, origin: int, size: int, overlap: int, out: list) -> None:
    """chunking.py:68-168 (_recursive_split)."""
    if not body:                                   # :75-76
        return
    if len(body) <= size:                          # :79-90
        _append(out, body, origin, origin + len(body))
        return
    chosen = next((s for s in seps if s in body), "")   # :93-97 ("" is in every string)
    if chosen == "":                               # :99-102
        _windows(body, origin, size, overlap, out)
        return
    pieces = body.split(chosen)                    # :105
    acc, acc_origin = "", origin                   # :106-107
    walked = 0                                     # running value of the sum at :139-141
    for k, piece in enumerate(pieces):
        unit = piece + chosen if k < len(pieces) - 1 else piece   # :111
        if len(acc) + len(unit) <= size:           # :114-115
            acc += unit
        else:
            _append(out, acc, acc_origin, acc_origin + len(acc))   # :117-126
            if overlap > 0 and acc:                # :129-135
                tail = acc[-overlap:]
                acc = tail + unit
                # the reference evaluates len(current_chunk) AFTER the reassignment: the terms cancel
                acc_origin = acc_origin + len(acc) - len(tail) - len(unit)
            else:                                  # :136-141
                acc = unit
                acc_origin = origin + walked
            if len(unit) > size:                   # :144-155
                level = seps.index(chosen)
                if level < len(seps) - 1:
                    _descend(unit, seps[level + 1:], acc_origin, size, overlap, out)
                    acc = ""
        walked += len(piece) + len(chosen)
    _append(out, acc, acc_origin, acc_origin + len(acc))           # :158-168


def _by_sentence(body: str, size: int, out: list) -> None:
    """chunking.py:193-239 (_sentence_chunk). The overlap plays no part in this strategy."""
    acc, acc_origin, cursor = "", 0, 0
    for raw in re.split(r"(?<=[.!?])\s+", body):   # :196-197
        sent = raw.strip()                          # :205-207
        if not sent:
            continue
        if len(acc) + len(sent) + 1 <= size:        # :209-214
            if acc:
                acc += " " + sent
            else:
                acc, acc_origin = sent, cursor
        else:                                       # :215-226
            if acc:
                out.append((acc, len(out), acc_origin, acc_origin + len(acc)))
            acc, acc_origin = sent, cursor
        cursor = body.find(sent, cursor) + len(sent)   # :228
    if acc:                                         # :230-238
        out.append((acc, len(out), acc_origin, acc_origin + len(acc)))


def chunk_text(text: str, chunk_size: int | None = None, chunk_overlap: int | None = None,
               strategy: str | None = None) -> list[tuple[str, int, int, int]]:
    """ChunkingService(chunk_size, chunk_overlap, strategy).chunk_text(text): chunking.py:22-45.
    A falsy argument (None or 0) takes the configured default, as ``x or settings.x`` does."""
    size = chunk_size or DEFAULTS["chunk_size"]
    overlap = chunk_overlap or DEFAULTS["chunk_overlap"]
    how = strategy or DEFAULTS["strategy"]
    out: list = []
    if not text or not text.strip():                # :35-36
        return out
    if how == "sentence":
        _by_sentence(text, size, out)
    elif how == "fixed":
        _windows(text, 0, size, overlap, out)       # :241-246
    else:                                           # "recursive" and anything else (:38-45)
        _descend(text, SEPARATORS, 0, size, overlap, out)
    return out
