// Host text side of the BM25 sparse model: what fastembed's Bm25 does to a string before any
// arithmetic (reference call sites: src/voitta/services/sparse_embedding.py:35,49;
// scripts/build_sparse_vectors.py:170; fastembed is un-vendored and unpinned, pyproject.toml:32;
// behaviour restated from its published implementation [EXT]):
//   remove_non_alphanumeric  re.sub(r"[^\w\s]", " ", text)
//   SimpleTokenizer          re.sub(r"[^\w]", " ", text.lower()) ; split on whitespace
//   _stem                    drop "_" (the only \w character of category P*), English stop-words,
//                            tokens longer than 40 code points ; Snowball English (Porter2) stem
//   token id                 abs(murmur3_x86_32(utf-8 stem, seed 0) as int32)
// String work stays on the host (it is branchy byte processing, ~100 tokens per chunk); the ids
// go to the GPU where bm25_tf_kernel counts and weights them.
// Unicode classes and lower-casing come from unicode_tables.inc (generated from CPython's
// unicodedata so that \w, \s and str.lower() agree with the Python calls). One deviation:
// str.lower()'s context rule for a word-final capital sigma is approximated (final when the
// previous code point is a word character and the next is not).

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

#include "engine_internal.h"
#include "host_parallel.h"
#include "unicode_tables.inc"

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

inline bool is_word(uint32_t cp) {
  if (cp < 128) return (cp >= '0' && cp <= '9') || (cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z') || cp == '_';
  return in_ranges(kWordRanges, cp);
}
inline bool is_space(uint32_t cp) {
  if (cp < 128) return cp == ' ' || (cp >= 9 && cp <= 13) || (cp >= 0x1C && cp <= 0x1F);
  return in_ranges(kSpaceRanges, cp);
}

// appends str.lower() of one code point
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

// lenient UTF-8 decoder (invalid bytes become U+FFFD, which is neither \w nor \s)
void decode_utf8(const char* s, size_t n, u32s* out) {
  out->clear();
  size_t i = 0;
  while (i < n) {
    unsigned char c = static_cast<unsigned char>(s[i]);
    uint32_t cp = 0xFFFD;
    int len = 1;
    if (c < 0x80) cp = c;
    else if ((c >> 5) == 6 && i + 1 < n) { cp = ((c & 0x1F) << 6) | (s[i + 1] & 0x3F); len = 2; }
    else if ((c >> 4) == 14 && i + 2 < n) { cp = ((c & 0x0F) << 12) | ((s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F); len = 3; }
    else if ((c >> 3) == 30 && i + 3 < n) {
      cp = ((c & 0x07) << 18) | ((s[i + 1] & 0x3F) << 12) | ((s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F);
      len = 4;
    }
    out->push_back(cp);
    i += len;
  }
}

void encode_utf8(const u32s& w, std::string* out) {
  out->clear();
  for (uint32_t cp : w) {
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
}

uint32_t murmur3_32(const uint8_t* data, size_t len, uint32_t seed) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h = seed;
  const size_t nblocks = len / 4;
  for (size_t i = 0; i < nblocks; ++i) {
    uint32_t k;
    memcpy(&k, data + 4 * i, 4);
    k *= c1; k = (k << 15) | (k >> 17); k *= c2;
    h ^= k; h = (h << 13) | (h >> 19); h = h * 5 + 0xe6546b64u;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k = 0;
  switch (len & 3) {
    case 3: k ^= static_cast<uint32_t>(tail[2]) << 16; [[fallthrough]];
    case 2: k ^= static_cast<uint32_t>(tail[1]) << 8; [[fallthrough]];
    case 1: k ^= tail[0]; k *= c1; k = (k << 15) | (k >> 17); k *= c2; h ^= k;
  }
  h ^= static_cast<uint32_t>(len);
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// ---- Snowball English (Porter2), restated from the published algorithm ---------------------------

inline bool vowel(char32_t c) { return c == U'a' || c == U'e' || c == U'i' || c == U'o' || c == U'u' || c == U'y'; }

bool ends(const u32s& w, const char* suf) {
  size_t n = strlen(suf);
  if (w.size() < n) return false;
  for (size_t i = 0; i < n; ++i)
    if (w[w.size() - n + i] != static_cast<char32_t>(static_cast<unsigned char>(suf[i]))) return false;
  return true;
}
bool equals(const u32s& w, const char* s) { return w.size() == strlen(s) && ends(w, s); }
void replace_end(u32s* w, size_t cut, const char* rep) {
  w->resize(w->size() - cut);
  for (const char* p = rep; *p; ++p) w->push_back(static_cast<char32_t>(static_cast<unsigned char>(*p)));
}
bool has_vowel(const u32s& w, size_t upto) {
  for (size_t i = 0; i < upto && i < w.size(); ++i)
    if (vowel(w[i])) return true;
  return false;
}
bool ends_short_syllable(const u32s& w, size_t n) {  // over w[0..n)
  if (n >= 3) {
    char32_t a = w[n - 3], b = w[n - 2], c = w[n - 1];
    return !vowel(a) && vowel(b) && !vowel(c) && c != U'w' && c != U'x' && c != U'Y';
  }
  return n == 2 && vowel(w[0]) && !vowel(w[1]);
}

struct Rule { const char* suf; const char* rep; };

void porter2(u32s* word) {
  u32s& w = *word;
  if (w.size() <= 2) return;
  static const Rule exc1[] = {{"skis", "ski"}, {"skies", "sky"}, {"dying", "die"}, {"lying", "lie"}, {"tying", "tie"},
                              {"idly", "idl"}, {"gently", "gentl"}, {"ugly", "ugli"}, {"early", "earli"},
                              {"only", "onli"}, {"singly", "singl"}, {"sky", "sky"}, {"news", "news"},
                              {"howe", "howe"}, {"atlas", "atlas"}, {"cosmos", "cosmos"}, {"bias", "bias"},
                              {"andes", "andes"}};
  for (const Rule& r : exc1)
    if (equals(w, r.suf)) { replace_end(&w, w.size(), r.rep); return; }
  if (w[0] == U'\'') w.erase(0, 1);
  if (!w.empty() && w[0] == U'y') w[0] = U'Y';
  for (size_t i = 1; i < w.size(); ++i)
    if (w[i] == U'y' && vowel(w[i - 1])) w[i] = U'Y';
  // regions
  size_t r1 = w.size();
  bool special = false;
  for (const char* p : {"gener", "commun", "arsen"}) {
    size_t n = strlen(p);
    if (w.size() >= n) {
      bool m = true;
      for (size_t i = 0; i < n; ++i) m = m && w[i] == static_cast<char32_t>(static_cast<unsigned char>(p[i]));
      if (m) { r1 = n; special = true; break; }
    }
  }
  if (!special)
    for (size_t i = 1; i < w.size(); ++i)
      if (!vowel(w[i]) && vowel(w[i - 1])) { r1 = i + 1; break; }
  size_t r2 = w.size();
  for (size_t i = r1 + 1; i < w.size(); ++i)
    if (!vowel(w[i]) && vowel(w[i - 1])) { r2 = i + 1; break; }
  // step 0
  if (ends(w, "'s'")) w.resize(w.size() - 3);
  else if (ends(w, "'s")) w.resize(w.size() - 2);
  else if (ends(w, "'")) w.resize(w.size() - 1);
  // step 1a
  if (ends(w, "sses")) w.resize(w.size() - 2);
  else if (ends(w, "ied") || ends(w, "ies")) replace_end(&w, 3, w.size() > 4 ? "i" : "ie");
  else if (ends(w, "us") || ends(w, "ss")) {}
  else if (ends(w, "s")) { if (w.size() >= 2 && has_vowel(w, w.size() - 2)) w.resize(w.size() - 1); }
  auto unmark = [&]() { for (auto& c : w) if (c == U'Y') c = U'y'; };
  for (const char* p : {"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed"})
    if (equals(w, p)) { unmark(); return; }
  // step 1b
  for (const char* suf : {"eedly", "ingly", "edly", "eed", "ing", "ed"}) {
    if (!ends(w, suf)) continue;
    size_t n = strlen(suf);
    if (!strcmp(suf, "eed") || !strcmp(suf, "eedly")) {
      if (w.size() - n >= r1) replace_end(&w, n, "ee");
    } else if (has_vowel(w, w.size() - n)) {
      w.resize(w.size() - n);
      if (ends(w, "at") || ends(w, "bl") || ends(w, "iz")) w.push_back(U'e');
      else if (ends(w, "bb") || ends(w, "dd") || ends(w, "ff") || ends(w, "gg") || ends(w, "mm") || ends(w, "nn") ||
               ends(w, "pp") || ends(w, "rr") || ends(w, "tt")) w.resize(w.size() - 1);
      else if (r1 >= w.size() && ends_short_syllable(w, w.size())) w.push_back(U'e');
    }
    break;
  }
  // step 1c
  if (w.size() > 2 && (w.back() == U'y' || w.back() == U'Y') && !vowel(w[w.size() - 2])) w.back() = U'i';
  // step 2
  static const Rule s2[] = {{"ization", "ize"}, {"ational", "ate"}, {"fulness", "ful"}, {"ousness", "ous"},
                            {"iveness", "ive"}, {"tional", "tion"}, {"biliti", "ble"}, {"lessli", "less"},
                            {"entli", "ent"}, {"ation", "ate"}, {"alism", "al"}, {"aliti", "al"}, {"ousli", "ous"},
                            {"iviti", "ive"}, {"fulli", "ful"}, {"enci", "ence"}, {"anci", "ance"}, {"abli", "able"},
                            {"izer", "ize"}, {"ator", "ate"}, {"alli", "al"}, {"bli", "ble"}, {"ogi", nullptr},
                            {"li", nullptr}};
  for (const Rule& r : s2) {
    if (!ends(w, r.suf)) continue;
    size_t n = strlen(r.suf), pos = w.size() - n;
    if (pos >= r1) {
      if (!strcmp(r.suf, "ogi")) { if (pos > 0 && w[pos - 1] == U'l') replace_end(&w, n, "og"); }
      else if (!strcmp(r.suf, "li")) {
        if (pos > 0 && strchr("cdeghkmnrt", static_cast<int>(w[pos - 1] < 128 ? w[pos - 1] : 0)) && w[pos - 1] < 128)
          w.resize(pos);
      } else replace_end(&w, n, r.rep);
    }
    break;
  }
  // step 3
  static const Rule s3[] = {{"ational", "ate"}, {"tional", "tion"}, {"alize", "al"}, {"icate", "ic"}, {"iciti", "ic"},
                            {"ative", nullptr}, {"ical", "ic"}, {"ness", ""}, {"ful", ""}};
  for (const Rule& r : s3) {
    if (!ends(w, r.suf)) continue;
    size_t n = strlen(r.suf), pos = w.size() - n;
    if (pos >= r1) {
      if (!r.rep) { if (pos >= r2) w.resize(pos); }
      else replace_end(&w, n, r.rep);
    }
    break;
  }
  // step 4
  for (const char* suf : {"ement", "ance", "ence", "able", "ible", "ment", "ant", "ent", "ism", "ate", "iti", "ous",
                          "ive", "ize", "ion", "al", "er", "ic"}) {
    if (!ends(w, suf)) continue;
    size_t n = strlen(suf), pos = w.size() - n;
    if (pos >= r2) {
      if (!strcmp(suf, "ion")) { if (pos > 0 && (w[pos - 1] == U's' || w[pos - 1] == U't')) w.resize(pos); }
      else w.resize(pos);
    }
    break;
  }
  // step 5
  if (ends(w, "e")) {
    size_t pos = w.size() - 1;
    if (pos >= r2 || (pos >= r1 && !ends_short_syllable(w, pos))) w.resize(pos);
  } else if (ends(w, "l")) {
    if (w.size() - 1 >= r2 && w.size() >= 2 && w[w.size() - 2] == U'l') w.resize(w.size() - 1);
  }
  unmark();
}

const std::unordered_set<std::string>& stopwords() {
  // the 179-entry NLTK English list shipped as english.txt with the Qdrant/bm25 model [EXT]
  static const std::unordered_set<std::string> s = {
      "i", "me", "my", "myself", "we", "our", "ours", "ourselves", "you", "you're", "you've", "you'll", "you'd",
      "your", "yours", "yourself", "yourselves", "he", "him", "his", "himself", "she", "she's", "her", "hers",
      "herself", "it", "it's", "its", "itself", "they", "them", "their", "theirs", "themselves", "what", "which",
      "who", "whom", "this", "that", "that'll", "these", "those", "am", "is", "are", "was", "were", "be", "been",
      "being", "have", "has", "had", "having", "do", "does", "did", "doing", "a", "an", "the", "and", "but", "if",
      "or", "because", "as", "until", "while", "of", "at", "by", "for", "with", "about", "against", "between",
      "into", "through", "during", "before", "after", "above", "below", "to", "from", "up", "down", "in", "out",
      "on", "off", "over", "under", "again", "further", "then", "once", "here", "there", "when", "where", "why",
      "how", "all", "any", "both", "each", "few", "more", "most", "other", "some", "such", "no", "nor", "not",
      "only", "own", "same", "so", "than", "too", "very", "s", "t", "can", "will", "just", "don", "don't", "should",
      "should've", "now", "d", "ll", "m", "o", "re", "ve", "y", "ain", "aren", "aren't", "couldn", "couldn't",
      "didn", "didn't", "doesn", "doesn't", "hadn", "hadn't", "hasn", "hasn't", "haven", "haven't", "isn", "isn't",
      "ma", "mightn", "mightn't", "mustn", "mustn't", "needn", "needn't", "shan", "shan't", "shouldn", "shouldn't",
      "wasn", "wasn't", "weren", "weren't", "won", "won't", "wouldn", "wouldn't"};
  return s;
}

constexpr size_t kTokenMaxLength = 40;

// text -> hashed stems in text order
void hashed_stems(const char* text, size_t n, std::vector<int32_t>* out) {
  u32s cps, low, tok;
  decode_utf8(text, n, &cps);
  // remove_non_alphanumeric + lower(): a code point that is neither \w nor \s becomes a space
  low.clear();
  for (size_t i = 0; i < cps.size(); ++i) {
    uint32_t cp = cps[i];
    if (!is_word(cp) && !is_space(cp)) { low.push_back(U' '); continue; }
    if (cp == 0x3A3) {  // capital sigma: final form when it ends a word
      bool prev_word = i > 0 && is_word(cps[i - 1]);
      bool next_word = i + 1 < cps.size() && is_word(cps[i + 1]);
      low.push_back(prev_word && !next_word ? 0x3C2 : 0x3C3);
      continue;
    }
    lower_cp(cp, &low);
  }
  std::string utf8;
  size_t i = 0;
  while (i <= low.size()) {
    bool brk = i == low.size() || !is_word(low[i]);  // [^\w] -> " " then split
    if (!brk) { tok.push_back(low[i]); ++i; continue; }
    if (!tok.empty()) {
      bool drop = (tok.size() == 1 && tok[0] == U'_') || tok.size() > kTokenMaxLength;
      if (!drop) {
        encode_utf8(tok, &utf8);
        drop = stopwords().count(utf8) != 0;
      }
      if (!drop) {
        porter2(&tok);
        if (!tok.empty()) {
          encode_utf8(tok, &utf8);
          uint32_t h = murmur3_32(reinterpret_cast<const uint8_t*>(utf8.data()), utf8.size(), 0);
          int32_t sgn = static_cast<int32_t>(h);
          // abs() of INT32_MIN does not fit an int32 (fastembed would overflow there too): clamp
          out->push_back(sgn == INT32_MIN ? INT32_MAX : (sgn < 0 ? -sgn : sgn));
        }
      }
      tok.clear();
    }
    ++i;
  }
}

}  // namespace

extern "C" {

// texts[i] has lens[i] bytes of UTF-8. Writes out_off[n+1] and up to cap ids; *out_needed = total
// ids (call again with a larger buffer when it exceeds cap). Host only, no engine needed.
int vr_bm25_tokenize(const char* const* texts, const int64_t* lens, int64_t n, int64_t* out_off,
                     int32_t* out_ids, int64_t cap, int64_t* out_needed) {
  VR_CHECK(n >= 0 && out_off && out_needed && (n == 0 || (texts && lens)), "bad arguments");
  // texts are independent: all host threads, then the ids in order (a buffer that is too small gets
  // the leading ids that fit, as before)
  std::vector<std::vector<int32_t>> per_text(static_cast<size_t>(n));
  vr::parallel_for(n, 64, [&](int64_t i) {
    hashed_stems(texts[i], static_cast<size_t>(lens[i]), &per_text[static_cast<size_t>(i)]);
  });
  int64_t total = 0;
  out_off[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    for (int32_t v : per_text[static_cast<size_t>(i)]) {
      if (total < cap && out_ids) out_ids[total] = v;
      ++total;
    }
    out_off[i + 1] = total;
  }
  *out_needed = total;
  if (total > cap) {  // offsets and *out_needed are valid; call again with a buffer of that many ids
    vr::set_error("vr_bm25_tokenize: %lld ids, buffer holds %lld", static_cast<long long>(total), static_cast<long long>(cap));
    return -2;
  }
  return 0;
}

// Snowball English stem of one lower-case UTF-8 word (exposed for the known-answer tests)
int vr_porter2_stem(const char* word, int64_t len, char* out, int64_t cap) {
  VR_CHECK(word && out && cap > 0, "bad arguments");
  u32s w;
  decode_utf8(word, static_cast<size_t>(len), &w);
  porter2(&w);
  std::string s;
  encode_utf8(w, &s);
  VR_CHECK(static_cast<int64_t>(s.size()) < cap, "output buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return 0;
}

}  // extern "C"
