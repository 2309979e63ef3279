// BERT WordPiece tokenizer behind the C-ABI (vr_wordpiece_*): the tokenise step of
// SentenceTransformer.encode (reference: src/voitta/services/embedding.py:40,68-73 -> [EXT]
// sentence-transformers -> HF `tokenizers`: BertNormalizer + BertPreTokenizer + WordPiece +
// "[CLS] $A [SEP]" template + right truncation). Host code: string
// work does not belong on the GPU; it runs beside the previous batch's forward pass.
//
// Pipeline, as the HF implementation defines it [EXT]:
//   clean_text        drop U+0000, U+FFFD and every Cc/Cf/Co character except \t \n \r; map
//                     White_Space characters (and \t \n \r) to ' '
//   chinese chars     ' ' before and after every CJK ideograph (fixed block list)
//   strip accents     NFD, then drop Mn   (on when lowercase is on, unless overridden)
//   lowercase         per-character Unicode lowercase (no final-sigma context rule)
//   pre-tokenise      split at White_Space (dropped) and at punctuation (ASCII punctuation or
//                     general category P*; every punctuation character is its own word)
//   WordPiece         greedy longest match, continuation prefix "##"; a word of more than 100
//                     characters, or with an unmatched remainder, becomes [UNK]
//   post              [CLS] ids... [SEP], ids truncated on the right to max_len - 2
// Pinned against the HF `tokenizers` library itself on synthetic vocabularies and adversarial
// Unicode text (tests/test_wordpiece_cpu.py).

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/voitta_engine.h"
#include "engine_internal.h"
#include "host_parallel.h"
#include "unicode_tables.inc"
#include "wordpiece_tables.inc"

struct vr_wordpiece {
  std::unordered_map<std::string, int32_t> vocab;
  int32_t unk = -1, cls = -1, sep = -1;
  bool lowercase = true, strip_accents = true, chinese = true, clean = true;
};

namespace {

using u32s = std::u32string;

template <size_t N>
bool in_ranges(const uint32_t (&r)[N][2], uint32_t cp) {
  size_t lo = 0, hi = N;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (cp < r[mid][0]) hi = mid;
    else if (cp > r[mid][1]) lo = mid + 1;
    else return true;
  }
  return false;
}

inline bool is_white_space(uint32_t c) {  // Unicode White_Space (Rust char::is_whitespace)
  return (c >= 9 && c <= 13) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
         c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
inline bool is_other(uint32_t c) { return in_ranges(kOtherRanges, c); }  // Cc | Cf | Co
inline bool is_punct(uint32_t c) {
  if (c < 128) return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
  return in_ranges(kPunctRanges, c);
}
inline bool is_cjk(uint32_t c) {
  return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) ||
         (c >= 0x2A700 && c <= 0x2B73F) || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B920 && c <= 0x2CEAF) ||
         (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}

inline uint32_t ccc_of(uint32_t c) {
  if (c < 0x300) return 0;
  const size_t n = sizeof(kCcc) / sizeof(kCcc[0]);
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (c < kCcc[mid].lo) hi = mid;
    else if (c > kCcc[mid].hi) lo = mid + 1;
    else return kCcc[mid].ccc;
  }
  return 0;
}

// canonical decomposition of one code point (fully expanded; Hangul syllables algorithmically)
inline void decompose(uint32_t c, u32s* out) {
  if (c < 0xC0) {
    out->push_back(c);
    return;
  }
  if (c >= 0xAC00 && c <= 0xD7A3) {
    const uint32_t s = c - 0xAC00;
    out->push_back(0x1100 + s / 588);
    out->push_back(0x1161 + (s % 588) / 28);
    if (s % 28) out->push_back(0x11A7 + s % 28);
    return;
  }
  const size_t n = sizeof(kDecomp) / sizeof(kDecomp[0]);
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (kDecomp[mid].cp < c) lo = mid + 1; else hi = mid;
  }
  if (lo < n && kDecomp[lo].cp == c) {
    for (uint32_t t : kDecomp[lo].to)
      if (t) out->push_back(t);
  } else {
    out->push_back(c);
  }
}

// NFD: decompose, then order every run of non-starters by canonical combining class (stable)
void nfd(const u32s& in, u32s* out) {
  out->clear();
  for (uint32_t c : in) decompose(c, out);
  size_t i = 0;
  const size_t n = out->size();
  while (i < n) {
    if (ccc_of((*out)[i]) == 0) {
      ++i;
      continue;
    }
    size_t j = i;
    while (j < n && ccc_of((*out)[j]) != 0) ++j;
    std::stable_sort(out->begin() + static_cast<std::ptrdiff_t>(i), out->begin() + static_cast<std::ptrdiff_t>(j),
                     [](char32_t a, char32_t b) { return ccc_of(a) < ccc_of(b); });
    i = j;
  }
}

inline void lower_cp(uint32_t cp, u32s* out) {
  if (cp < 128) {
    out->push_back((cp >= 'A' && cp <= 'Z') ? cp + 32 : cp);
    return;
  }
  const size_t n = sizeof(kLower) / sizeof(kLower[0]);
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (kLower[mid].cp < cp) lo = mid + 1; else hi = mid;
  }
  if (lo < n && kLower[lo].cp == cp) {
    for (uint32_t t : kLower[lo].to)
      if (t) out->push_back(t);
  } else {
    out->push_back(cp);
  }
}

// strict-enough UTF-8 decoder: malformed bytes become U+FFFD (which clean_text then removes,
// like a Python str that was decoded with errors="replace")
void decode_utf8(const char* s, size_t n, u32s* out) {
  out->clear();
  size_t i = 0;
  while (i < n) {
    const unsigned char c = static_cast<unsigned char>(s[i]);
    uint32_t cp = 0xFFFD;
    int len = 1;
    auto cont = [&](size_t k) { return i + k < n && (static_cast<unsigned char>(s[i + k]) & 0xC0) == 0x80; };
    if (c < 0x80) cp = c;
    else if ((c >> 5) == 6 && cont(1)) { cp = ((c & 0x1Fu) << 6) | (s[i + 1] & 0x3Fu); len = 2; }
    else if ((c >> 4) == 14 && cont(1) && cont(2)) { cp = ((c & 0x0Fu) << 12) | ((s[i + 1] & 0x3Fu) << 6) | (s[i + 2] & 0x3Fu); len = 3; }
    else if ((c >> 3) == 30 && cont(1) && cont(2) && cont(3)) {
      cp = ((c & 0x07u) << 18) | ((s[i + 1] & 0x3Fu) << 12) | ((s[i + 2] & 0x3Fu) << 6) | (s[i + 3] & 0x3Fu);
      len = 4;
    }
    if (cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) cp = 0xFFFD;
    out->push_back(cp);
    i += static_cast<size_t>(len);
  }
}

inline void append_utf8(uint32_t cp, std::string* out) {
  if (cp < 0x80) out->push_back(static_cast<char>(cp));
  else if (cp < 0x800) { out->push_back(static_cast<char>(0xC0 | (cp >> 6))); out->push_back(static_cast<char>(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    out->push_back(static_cast<char>(0xE0 | (cp >> 12)));
    out->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
    out->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
  } else {
    out->push_back(static_cast<char>(0xF0 | (cp >> 18)));
    out->push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
    out->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
    out->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
  }
}

void normalize(const vr_wordpiece& t, const u32s& in, u32s* out) {
  u32s a, b;
  a.reserve(in.size() + 16);
  for (uint32_t c : in) {
    if (t.clean) {
      if (c == 0 || c == 0xFFFD) continue;
      const bool keep_ws = c == '\t' || c == '\n' || c == '\r';
      if (!keep_ws && is_other(c)) continue;
      if (keep_ws || is_white_space(c)) c = ' ';
    }
    if (t.chinese && is_cjk(c)) {
      a.push_back(' ');
      a.push_back(c);
      a.push_back(' ');
    } else {
      a.push_back(c);
    }
  }
  if (t.strip_accents) {
    nfd(a, &b);
    a.clear();
    for (uint32_t c : b)
      if (!in_ranges(kMnRanges, c)) a.push_back(c);
  }
  if (t.lowercase) {
    b.clear();
    for (uint32_t c : a) lower_cp(c, &b);
    a.swap(b);
  }
  out->swap(a);
}

// one pre-tokenised word -> ids
void wordpiece(const vr_wordpiece& t, const u32s& text, size_t w0, size_t w1, std::vector<int32_t>* ids,
               std::string* buf, std::vector<uint32_t>* pos) {
  if (w1 - w0 > 100) {  // max_input_chars_per_word
    ids->push_back(t.unk);
    return;
  }
  buf->clear();
  pos->clear();
  for (size_t i = w0; i < w1; ++i) {
    pos->push_back(static_cast<uint32_t>(buf->size()));
    append_utf8(text[i], buf);
  }
  pos->push_back(static_cast<uint32_t>(buf->size()));
  const size_t n = w1 - w0, first = ids->size();
  std::string piece;
  size_t start = 0;
  while (start < n) {
    size_t end = n;
    int32_t found = -1;
    while (end > start) {
      piece.assign(start ? "##" : "");
      piece.append(*buf, (*pos)[start], (*pos)[end] - (*pos)[start]);
      auto it = t.vocab.find(piece);
      if (it != t.vocab.end()) {
        found = it->second;
        break;
      }
      --end;
    }
    if (found < 0) {
      ids->resize(first);
      ids->push_back(t.unk);
      return;
    }
    ids->push_back(found);
    start = end;
  }
}

void encode_one(const vr_wordpiece& t, const char* s, size_t n, int32_t max_len, std::vector<int32_t>* ids) {
  u32s raw, text;
  decode_utf8(s, n, &raw);
  normalize(t, raw, &text);
  ids->clear();
  ids->push_back(t.cls);
  std::string buf;
  std::vector<uint32_t> pos;
  const size_t budget = max_len > 2 ? static_cast<size_t>(max_len) - 1 : 1;  // ids before [SEP]
  size_t i = 0;
  const size_t len = text.size();
  while (i < len && ids->size() < budget + 64) {  // (+64: a word may add several pieces; trimmed below)
    if (is_white_space(text[i])) {
      ++i;
      continue;
    }
    if (is_punct(text[i])) {
      wordpiece(t, text, i, i + 1, ids, &buf, &pos);
      ++i;
      continue;
    }
    size_t j = i;
    while (j < len && !is_white_space(text[j]) && !is_punct(text[j])) ++j;
    wordpiece(t, text, i, j, ids, &buf, &pos);
    i = j;
  }
  if (max_len >= 2 && ids->size() > budget) ids->resize(budget);
  ids->push_back(t.sep);
}

}  // namespace

extern "C" {

int vr_wordpiece_create(const char* const* vocab_tokens, int32_t n_vocab, int32_t lowercase, int32_t strip_accents,
                        int32_t handle_chinese_chars, int32_t clean_text, vr_wordpiece** out) {
  VR_CHECK(vocab_tokens && out && n_vocab > 0, "bad arguments");
  vr_wordpiece* t = new vr_wordpiece();
  t->vocab.reserve(static_cast<size_t>(n_vocab) * 2);
  // HF's WordPiece::read_file builds {token: line}: a later duplicate line overwrites an earlier one
  for (int32_t i = 0; i < n_vocab; ++i)
    if (vocab_tokens[i]) t->vocab[vocab_tokens[i]] = i;
  auto special = [&](const char* name) {
    auto it = t->vocab.find(name);
    return it == t->vocab.end() ? -1 : it->second;
  };
  t->unk = special("[UNK]");
  t->cls = special("[CLS]");
  t->sep = special("[SEP]");
  if (t->unk < 0 || t->cls < 0 || t->sep < 0) {
    delete t;
    vr::set_error("vocabulary lacks [UNK], [CLS] or [SEP]");
    return -1;
  }
  t->lowercase = lowercase != 0;
  t->strip_accents = strip_accents < 0 ? t->lowercase : strip_accents != 0;
  t->chinese = handle_chinese_chars != 0;
  t->clean = clean_text != 0;
  *out = t;
  return 0;
}

void vr_wordpiece_destroy(vr_wordpiece* t) { delete t; }

int vr_wordpiece_encode(const vr_wordpiece* t, const char* const* texts, const int64_t* text_lens, int64_t n_texts,
                        int32_t max_len, int64_t* out_offsets, int32_t* out_ids, int64_t capacity, int64_t* needed) {
  VR_CHECK(t && (n_texts == 0 || (texts && text_lens)) && out_offsets && needed, "bad arguments");
  VR_CHECK(max_len >= 2, "max_len %d cannot hold [CLS] and [SEP]", max_len);
  // texts are independent: tokenise them on all host threads, then lay the ids out in order
  std::vector<std::vector<int32_t>> per_text(static_cast<size_t>(n_texts));
  vr::parallel_for(n_texts, 64, [&](int64_t i) {
    encode_one(*t, texts[i], static_cast<size_t>(text_lens[i]), max_len, &per_text[static_cast<size_t>(i)]);
  });
  int64_t total = 0;
  out_offsets[0] = 0;
  for (int64_t i = 0; i < n_texts; ++i) {
    total += static_cast<int64_t>(per_text[static_cast<size_t>(i)].size());
    out_offsets[i + 1] = total;
  }
  if (out_ids && total <= capacity)
    vr::parallel_for(n_texts, 256, [&](int64_t i) {
      const std::vector<int32_t>& ids = per_text[static_cast<size_t>(i)];
      if (!ids.empty()) memcpy(out_ids + out_offsets[i], ids.data(), ids.size() * sizeof(int32_t));
    });
  *needed = total;
  if (total > capacity) {
    vr::set_error("output buffer holds %lld ids, %lld needed", static_cast<long long>(capacity), static_cast<long long>(total));
    return -2;  // offsets and *needed are valid: call again with a larger buffer
  }
  return 0;
}

}  // extern "C"
