"""BERT WordPiece tokenizer through the C-ABI (vr_wordpiece_*, csrc/wordpiece.cpp): the tokenise
step of SentenceTransformer.encode (reference: src/voitta/services/embedding.py:40,68-73 -> [EXT]
HF tokenizers). Returns packed ids + offsets, the form vr_encode / vr_index_batch take."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from ._lib import check, load_library


class WordPieceTokenizer:
    def __init__(self, vocab: list[str], lowercase: bool = True, strip_accents: bool | None = None,
                 handle_chinese_chars: bool = True, clean_text: bool = True, max_length: int = 512):
        self._lib = load_library()
        self.vocab_size = len(vocab)
        self.max_length = int(max_length)
        raw = [t.encode("utf-8") for t in vocab]
        arr = (C.c_char_p * len(raw))(*raw)
        h = C.c_void_p()
        check(self._lib.vr_wordpiece_create(arr, len(raw), int(lowercase), -1 if strip_accents is None else int(strip_accents),
                                            int(handle_chinese_chars), int(clean_text), C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vr_wordpiece_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_pretrained(cls, path: str, max_length: int = 512) -> "WordPieceTokenizer":
        """vocab.txt (+ tokenizer_config.json) or tokenizer.json of a BERT checkpoint directory."""
        lowercase, strip, chinese, clean = True, None, True, True
        cfg_path = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(cfg_path):
            cfg = json.load(open(cfg_path, encoding="utf-8"))
            lowercase = bool(cfg.get("do_lower_case", True))
            strip = cfg.get("strip_accents")
            chinese = bool(cfg.get("tokenize_chinese_chars", True))
        vocab_path = os.path.join(path, "vocab.txt")
        if os.path.exists(vocab_path):
            vocab = [line.rstrip("\n") for line in open(vocab_path, encoding="utf-8")]
        else:
            tj = json.load(open(os.path.join(path, "tokenizer.json"), encoding="utf-8"))
            if tj["model"]["type"] != "WordPiece":
                raise ValueError(f"tokenizer model {tj['model']['type']} is not WordPiece")
            vocab = [""] * (max(tj["model"]["vocab"].values()) + 1)
            for tok, i in tj["model"]["vocab"].items():
                vocab[i] = tok
            norm = tj.get("normalizer") or {}
            if norm.get("type") == "BertNormalizer":
                lowercase = bool(norm.get("lowercase", True))
                strip = norm.get("strip_accents")
                chinese = bool(norm.get("handle_chinese_chars", True))
                clean = bool(norm.get("clean_text", True))
        return cls(vocab, lowercase, strip, chinese, clean, max_length)

    def encode_batch(self, texts: list[str]) -> tuple[np.ndarray, np.ndarray]:
        """-> (ids int32[total], offsets int64[n + 1]); every sequence is [CLS] ... [SEP]."""
        n = len(texts)
        raw = [t.encode("utf-8", "replace") for t in texts]
        arr = (C.c_char_p * max(n, 1))(*raw)
        lens = np.asarray([len(b) for b in raw], np.int64)
        off = np.zeros(n + 1, np.int64)
        cap = int(sum(min(len(b) + 2, self.max_length) for b in raw)) + 2  # a piece consumes >= 1 byte
        ids = np.empty(cap, np.int32)
        needed = C.c_int64()
        rc = self._lib.vr_wordpiece_encode(self._h, arr, lens.ctypes.data_as(C.POINTER(C.c_int64)), n, self.max_length,
                                           off.ctypes.data_as(C.POINTER(C.c_int64)), ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                           cap, C.byref(needed))
        check(rc)
        return ids[: int(needed.value)].copy(), off
