"""TEST INFRASTRUCTURE — Python restatement of fastembed's ``Qdrant/bm25`` sparse model, the
un-vendored dependency behind SparseEmbeddingService (reference call sites:
src/voitta/services/sparse_embedding.py:25,35,49; scripts/build_sparse_vectors.py:124,170;
pyproject.toml:32 ``fastembed>=0.7.0``, not pinned, absent from this container).

Restated from the published behaviour of fastembed.sparse.bm25.Bm25 [EXT]:
  remove_non_alphanumeric : re.sub(r"[^\\w\\s]", " ", text)
  SimpleTokenizer         : re.sub(r"[^\\w]", " ", text.lower()) ; collapse whitespace ; split
  _stem                   : drop single-character punctuation tokens (Unicode category P*), English
                            stop-words, tokens longer than 40 characters ; Snowball English stem ;
                            drop empty stems
  token id                : abs(murmur3_x86_32(stem utf-8, seed 0) as int32)
  document weight         : tf = c*(k+1) / (c + k*(1 - b + b*doc_len/avg_len)), k=1.2 b=0.75 avg_len=256,
                            doc_len = number of stemmed tokens, Python floats (f64)
  query                   : set of token ids, all values 1.0
Stop-words: the 179-entry NLTK English list the model repository ships as english.txt [EXT].
Stemmer: the Snowball "english" (Porter2) algorithm, restated from its published definition.

PARITY UNPINNED: neither fastembed nor py_rust_stemmers nor mmh3 is installable here and the
reference's tests hold no BM25 fixture. Pinned only by tests/golden/bm25_kat.json: published
Snowball vocabulary pairs, murmur3 reference values and hand-derived tf values.
"""
from __future__ import annotations

import re
import sys
import unicodedata

K, B, AVG_LEN = 1.2, 0.75, 256.0
TOKEN_MAX_LENGTH = 40

STOPWORDS = frozenset("""i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself
yourselves he him his himself she she's her hers herself it it's its itself they them their theirs themselves what
which who whom this that that'll these those am is are was were be been being have has had having do does did doing a
an the and but if or because as until while of at by for with about against between into through during before after
above below to from up down in out on off over under again further then once here there when where why how all any
both each few more most other some such no nor not only own same so than too very s t can will just don don't should
should've now d ll m o re ve y ain aren aren't couldn couldn't didn didn't doesn doesn't hadn hadn't hasn hasn't haven
haven't isn isn't ma mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't wasn wasn't weren
weren't won won't wouldn wouldn't""".split())
assert len(STOPWORDS) == 179

_PUNCT = None


def punctuation() -> frozenset:
    global _PUNCT
    if _PUNCT is None:
        _PUNCT = frozenset(chr(i) for i in range(sys.maxunicode + 1) if unicodedata.category(chr(i)).startswith("P"))
    return _PUNCT


# ---- murmur3 -----------------------------------------------------------------------------------

def murmur3_32(data: bytes, seed: int = 0) -> int:
    """MurmurHash3_x86_32, unsigned."""
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed & 0xFFFFFFFF
    n = len(data)
    for i in range(0, n - n % 4, 4):
        k = int.from_bytes(data[i:i + 4], "little")
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
        h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
    tail = data[n - n % 4:]
    k = 0
    if len(tail) >= 3:
        k ^= tail[2] << 16
    if len(tail) >= 2:
        k ^= tail[1] << 8
    if len(tail) >= 1:
        k ^= tail[0]
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
    h ^= n
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def token_id(stem: str) -> int:
    h = murmur3_32(stem.encode("utf-8"), 0)
    signed = h - (1 << 32) if h & 0x80000000 else h  # mmh3.hash returns a signed int32
    return abs(signed)


# ---- Snowball English (Porter2) ------------------------------------------------------------------

_V = "aeiouy"
_DOUBLE = ("bb", "dd", "ff", "gg", "mm", "nn", "pp", "rr", "tt")
_LI = "cdeghkmnrt"
_EXC1 = {"skis": "ski", "skies": "sky", "dying": "die", "lying": "lie", "tying": "tie", "idly": "idl",
         "gently": "gentl", "ugly": "ugli", "early": "earli", "only": "onli", "singly": "singl",
         "sky": "sky", "news": "news", "howe": "howe", "atlas": "atlas", "cosmos": "cosmos", "bias": "bias",
         "andes": "andes"}
_EXC2 = {"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed"}
_STEP2 = [("ization", "ize"), ("ational", "ate"), ("fulness", "ful"), ("ousness", "ous"), ("iveness", "ive"),
          ("tional", "tion"), ("biliti", "ble"), ("lessli", "less"), ("entli", "ent"), ("ation", "ate"),
          ("alism", "al"), ("aliti", "al"), ("ousli", "ous"), ("iviti", "ive"), ("fulli", "ful"), ("enci", "ence"),
          ("anci", "ance"), ("abli", "able"), ("izer", "ize"), ("ator", "ate"), ("alli", "al"), ("bli", "ble"),
          ("ogi", None), ("li", None)]
_STEP3 = [("ational", "ate"), ("tional", "tion"), ("alize", "al"), ("icate", "ic"), ("iciti", "ic"), ("ative", None),
          ("ical", "ic"), ("ness", ""), ("ful", "")]
_STEP4 = ["ement", "ance", "ence", "able", "ible", "ment", "ant", "ent", "ism", "ate", "iti", "ous", "ive", "ize",
          "ion", "al", "er", "ic"]


def _regions(w: str):
    r1 = len(w)
    for pref in ("gener", "commun", "arsen"):
        if w.startswith(pref):
            r1 = len(pref)
            break
    else:
        for i in range(1, len(w)):
            if w[i] not in _V and w[i - 1] in _V:
                r1 = i + 1
                break
    r2 = len(w)
    for i in range(r1 + 1, len(w)):
        if w[i] not in _V and w[i - 1] in _V:
            r2 = i + 1
            break
    return r1, r2


def _ends_short_syllable(w: str) -> bool:
    n = len(w)
    if n >= 3:
        return w[-3] not in _V and w[-2] in _V and w[-1] not in _V and w[-1] not in "wxY"
    return n == 2 and w[0] in _V and w[1] not in _V


def _has_vowel(s: str) -> bool:
    return any(c in _V for c in s)


def stem(word: str) -> str:
    w = word
    if len(w) <= 2:
        return w
    if w in _EXC1:
        return _EXC1[w]
    if w[0] == "'":
        w = w[1:]
    # mark consonant y
    chars = list(w)
    if chars and chars[0] == "y":
        chars[0] = "Y"
    for i in range(1, len(chars)):
        if chars[i] == "y" and chars[i - 1] in _V:
            chars[i] = "Y"
    w = "".join(chars)
    r1, r2 = _regions(w)
    # step 0
    for suf in ("'s'", "'s", "'"):
        if w.endswith(suf):
            w = w[: -len(suf)]
            break
    # step 1a
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ied") or w.endswith("ies"):
        w = w[:-3] + ("i" if len(w) > 4 else "ie")
    elif w.endswith("us") or w.endswith("ss"):
        pass
    elif w.endswith("s"):
        if _has_vowel(w[:-2]):
            w = w[:-1]
    if w in _EXC2:
        return w.replace("Y", "y")
    # step 1b
    for suf in ("eedly", "ingly", "edly", "eed", "ing", "ed"):
        if w.endswith(suf):
            if suf in ("eed", "eedly"):
                if len(w) - len(suf) >= r1:
                    w = w[: -len(suf)] + "ee"
            elif _has_vowel(w[: -len(suf)]):
                w = w[: -len(suf)]
                if w.endswith(("at", "bl", "iz")):
                    w += "e"
                elif w.endswith(_DOUBLE):
                    w = w[:-1]
                elif r1 >= len(w) and _ends_short_syllable(w):
                    w += "e"
            break
    # step 1c
    if len(w) > 2 and w[-1] in "yY" and w[-2] not in _V:
        w = w[:-1] + "i"
    # step 2
    for suf, rep in _STEP2:
        if w.endswith(suf):
            pos = len(w) - len(suf)
            if pos >= r1:
                if suf == "ogi":
                    if pos > 0 and w[pos - 1] == "l":
                        w = w[:pos] + "og"
                elif suf == "li":
                    if pos > 0 and w[pos - 1] in _LI:
                        w = w[:pos]
                else:
                    w = w[:pos] + rep
            break
    # step 3
    for suf, rep in _STEP3:
        if w.endswith(suf):
            pos = len(w) - len(suf)
            if pos >= r1:
                if suf == "ative":
                    if pos >= r2:
                        w = w[:pos]
                else:
                    w = w[:pos] + rep
            break
    # step 4
    for suf in _STEP4:
        if w.endswith(suf):
            pos = len(w) - len(suf)
            if pos >= r2:
                if suf == "ion":
                    if pos > 0 and w[pos - 1] in "st":
                        w = w[:pos]
                else:
                    w = w[:pos]
            break
    # step 5
    if w.endswith("e"):
        pos = len(w) - 1
        if pos >= r2 or (pos >= r1 and not _ends_short_syllable(w[:-1])):
            w = w[:-1]
    elif w.endswith("l"):
        if len(w) - 1 >= r2 and len(w) >= 2 and w[-2] == "l":
            w = w[:-1]
    return w.replace("Y", "y")


# ---- fastembed pipeline ---------------------------------------------------------------------------

def tokenize(text: str) -> list[str]:
    text = re.sub(r"[^\w\s]", " ", text, flags=re.UNICODE)  # remove_non_alphanumeric
    text = re.sub(r"[^\w]", " ", text.lower())               # SimpleTokenizer.tokenize
    text = re.sub(r"\s+", " ", text)
    return text.strip().split()


def stems(text: str) -> list[str]:
    out = []
    punct = punctuation()
    for token in tokenize(text):
        lower = token.lower()
        if token in punct:
            continue
        if lower in STOPWORDS:
            continue
        if len(token) > TOKEN_MAX_LENGTH:
            continue
        s = stem(lower)
        if s:
            out.append(s)
    return out


def term_frequency(stemmed: list[str], k: float = K, b: float = B, avg_len: float = AVG_LEN) -> dict[int, float]:
    """Bm25._term_frequency: dict insertion order = first occurrence of each stem."""
    counter: dict[str, int] = {}
    for s in stemmed:
        counter[s] = counter.get(s, 0) + 1
    doc_len = len(stemmed)
    tf_map: dict[int, float] = {}
    for s, num in counter.items():
        tid = token_id(s)
        tf_map[tid] = num * (k + 1)
        tf_map[tid] /= num + k * (1 - b + b * doc_len / avg_len)
    return tf_map


def embed(texts: list[str]):
    """SparseTextEmbedding.embed -> [(indices, values)] in fastembed's order (first occurrence)."""
    out = []
    for t in texts:
        m = term_frequency(stems(t))
        out.append((list(m.keys()), list(m.values())))
    return out


def query_embed(text: str):
    """Bm25.query_embed: the set of token ids, values all 1.0 (set order is unspecified; sorted here)."""
    ids = sorted({token_id(s) for s in stems(text)})
    return ids, [1.0] * len(ids)


def hashed_stems(text: str) -> list[int]:
    """The stream the device kernel consumes: abs(murmur3) of every stemmed token in text order."""
    return [token_id(s) for s in stems(text)]


def tf_from_hashed(ids: list[int], k: float = K, b: float = B, avg_len: float = AVG_LEN):
    """What vr_bm25_tf must return for one document: ascending ids, f64 weights. Colliding hashes
    of different stems are merged (documented deviation from term_frequency above)."""
    cnt: dict[int, int] = {}
    for t in ids:
        cnt[t] = cnt.get(t, 0) + 1
    doc_len = len(ids)
    idx = sorted(cnt)
    val = []
    for t in idx:
        num = cnt[t]
        v = num * (k + 1)
        v /= num + k * (1 - b + b * doc_len / avg_len)
        val.append(v)
    return idx, val
