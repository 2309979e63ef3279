"""vr_wordpiece_* (csrc/wordpiece.cpp) against the HF `tokenizers` library itself - the component
sentence-transformers uses for SentenceTransformer.encode's tokenise step (reference:
src/voitta/services/embedding.py:40,68-73 [EXT]; SURVEY.md section 8c lists this library as the pin
for the build's WordPiece). Synthetic vocabularies, adversarial Unicode text; ids must be identical."""
import numpy as np
import pytest

tokenizers = pytest.importorskip("tokenizers")


def _hf(vocab, lowercase=True, strip_accents=None, chinese=True, clean=True, max_length=64):
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    ids = {t: i for i, t in enumerate(vocab)}
    tok = Tokenizer(models.WordPiece(vocab=ids, unk_token="[UNK]", max_input_chars_per_word=100))
    tok.normalizer = normalizers.BertNormalizer(clean_text=clean, handle_chinese_chars=chinese,
                                                strip_accents=strip_accents, lowercase=lowercase)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]",
                                                       special_tokens=[("[CLS]", ids["[CLS]"]), ("[SEP]", ids["[SEP]"])])
    tok.enable_truncation(max_length=max_length)
    return tok


BLOCKS = [(0x20, 0x7E), (0x20, 0x7E), (0x20, 0x7E), (0xA0, 0xFF), (0x100, 0x17F), (0x180, 0x24F), (0x300, 0x36F),
          (0x370, 0x3FF), (0x400, 0x4FF), (0x590, 0x5FF), (0x600, 0x6FF), (0x900, 0x97F), (0xE00, 0xE7F), (0x1100, 0x11FF),
          (0x1E00, 0x1EFF), (0x1F00, 0x1FFF), (0x2000, 0x206F), (0x20A0, 0x20CF), (0x2100, 0x214F), (0x2190, 0x21FF),
          (0x3000, 0x303F), (0x3040, 0x30FF), (0x4E00, 0x4E80), (0xAC00, 0xAC80), (0xD700, 0xD7A3), (0xF900, 0xF940),
          (0xFB00, 0xFB4F), (0xFE50, 0xFE6F), (0xFF00, 0xFFEF), (0x1F600, 0x1F64F), (0x20000, 0x20040), (0x2F800, 0x2F820),
          (0x0, 0x1F), (0x7F, 0x9F), (0xE000, 0xE010), (0xFFF0, 0xFFFF), (0x1D400, 0x1D430), (0x10400, 0x1044F)]
SPECIALS = ["\t", "\n", "\r", " ", " ", "　", "", " ", "​", "﻿", "�", "­",
            "͸", "İ", "ǅ", "ß", "Σ", "ς", "ﬁ", "Å", "Å", "é", "é",
            "ạ̈", "ạ̈", "̈́", "각", "각"]


def _random_text(rng, n_chars):
    out = []
    for _ in range(n_chars):
        r = rng.random()
        if r < 0.12:
            out.append(" ")
        elif r < 0.2:
            out.append(SPECIALS[rng.integers(len(SPECIALS))])
        else:
            lo, hi = BLOCKS[rng.integers(len(BLOCKS))]
            cp = int(rng.integers(lo, hi + 1))
            if 0xD800 <= cp <= 0xDFFF:
                cp = 0x41
            out.append(chr(cp))
    return "".join(out)


def _vocab(rng, texts, tok_plain):
    """specials + single characters (bare and ##-continued) of what the normaliser produces for a
    sample of the texts + random multi-character pieces cut from them."""
    pieces = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = set()
    multi = set()
    for t in texts[::3]:
        norm = tok_plain.normalizer.normalize_str(t)
        for w in norm.split():
            chars.update(w)
            if len(w) >= 2 and rng.random() < 0.5:
                a = int(rng.integers(0, len(w) - 1))
                b = int(rng.integers(a + 1, min(len(w), a + 5) + 1))
                multi.add(w[a:b] if a == 0 else "##" + w[a:b])
    chars = sorted(chars - {"\x00"})  # the C-ABI takes NUL-terminated vocabulary entries (as vocab.txt lines are)
    keep = [c for c in chars if rng.random() < 0.8]          # 20 % of the characters stay out: [UNK] words
    pieces += keep + ["##" + c for c in keep if rng.random() < 0.85] + sorted(multi)
    pieces += ["hello", "world", "##ing", "##ed", "un", "##believ", "##able", "token", "##izer", "##s"]
    # the C-ABI takes NUL-terminated vocabulary entries (as vocab.txt lines are): no piece may hold U+0000
    return [p for p in dict.fromkeys(pieces) if "\x00" not in p]


@pytest.mark.parametrize("lowercase,strip,chinese,clean", [(True, None, True, True), (False, None, True, True),
                                                           (True, False, False, True), (False, True, True, False)])
def test_ids_equal_hf_tokenizers(lowercase, strip, chinese, clean):
    from voitta_rag_amd.wordpiece import WordPieceTokenizer

    rng = np.random.default_rng([int(lowercase), 2 if strip is None else int(strip), int(chinese), int(clean)])
    texts = [_random_text(rng, int(rng.integers(0, 120))) for _ in range(1500)]
    texts += ["", " ", "hello world", "unbelievable tokenizers!", "x" * 100 + " " + "y" * 101, "a" * 300,
              "Hello, World! It's 3.14... (really)", "中文字符 mixed 한국어 ＦＵＬＬ ｗｉｄｔｈ",
              "\x00�\x7f"]
    plain = _hf(["[UNK]", "[CLS]", "[SEP]"], lowercase, strip, chinese, clean)
    vocab = _vocab(rng, texts, plain)
    for max_len in (64, 16, 2):
        hf = _hf(vocab, lowercase, strip, chinese, clean, max_length=max_len)
        ours = WordPieceTokenizer(vocab, lowercase, strip, chinese, clean, max_length=max_len)
        ids, off = ours.encode_batch(texts)
        want = hf.encode_batch(texts)
        bad = 0
        for i, enc in enumerate(want):
            got = ids[off[i]:off[i + 1]].tolist()
            if got != enc.ids:
                bad += 1
                if bad <= 3:
                    print(f"text {i} {texts[i]!r}\n  normalised {hf.normalizer.normalize_str(texts[i])!r}\n"
                          f"  want {enc.tokens}\n  got  {[vocab[j] for j in got]}")
        assert bad == 0, f"{bad} of {len(texts)} texts differ (max_len {max_len})"
        ours.close()


def test_from_pretrained_and_errors(tmp_path):
    from voitta_rag_amd._lib import EngineError
    from voitta_rag_amd.wordpiece import WordPieceTokenizer

    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "hello", "world", "##s", "!", "he", "##llo"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    (tmp_path / "tokenizer_config.json").write_text('{"do_lower_case": true}')
    t = WordPieceTokenizer.from_pretrained(str(tmp_path), max_length=8)
    ids, off = t.encode_batch(["Hello worlds!", "HELLO " * 20])
    assert ids[off[0]:off[1]].tolist() == [2, 4, 5, 6, 7, 3]
    assert ids[off[1]:off[2]].tolist() == [2] + [4] * 6 + [3]          # truncated to 8 with specials
    hf = _hf(vocab, max_length=8)
    hf.save(str(tmp_path / "tokenizer.json"))
    (tmp_path / "vocab.txt").unlink()
    t2 = WordPieceTokenizer.from_pretrained(str(tmp_path), max_length=8)  # from tokenizer.json
    assert np.array_equal(t2.encode_batch(["Hello worlds!"])[0], ids[off[0]:off[1]])
    with pytest.raises(EngineError, match=r"\[UNK\]"):
        WordPieceTokenizer(["a", "b"])
