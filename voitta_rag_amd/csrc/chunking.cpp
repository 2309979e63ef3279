// Text chunker behind the C-ABI (vr_chunk_texts): the step in front of the indexing path, the
// reference's ChunkingService.chunk_text (src/voitta/services/chunking.py:33-241; callers
// services/indexing.py:380,515). Host code, one document per host thread.
//
// Everything is counted in CHARACTERS (Unicode code points, Python's len()), so documents are
// decoded to UTF-32 first; start/end offsets are code-point offsets exactly as the reference
// reports them, including the places where its bookkeeping is loose (see the notes below).
// The three strategies:
//   recursive (chunking.py:46-168)  split at the first separator of
//        "\n\n" "\n" ". " "? " "! " "; " ", " " " that occurs in the text, pack the parts greedily
//        up to chunk_size, carry the last chunk_overlap characters into the next chunk, descend to
//        the next separators for a part that is itself too long, and cut by size when none is left
//   sentence  (chunking.py:193-239) split after [.!?] at white-space runs, pack stripped sentences
//        joined by one blank
//   fixed     (chunking.py:170-191) windows of chunk_size every chunk_size - chunk_overlap
// White space is Python's str.isspace() set, which str.strip() and the regex class \s share.
// PARITY UNPINNED: the reference chunker cannot be imported in the build container (its config
// module needs python-dotenv, which is absent) and the reference holds no chunking fixtures, so
// the checker is oracle/chunking.py, a line-by-line restatement (tests/test_chunking_cpu.py).

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/voitta_engine.h"
#include "engine_internal.h"
#include "host_parallel.h"

namespace {

using Text = std::u32string;

struct Piece {          // one chunk of one document
  int64_t start, end;   // code-point offsets as the reference reports them
  Text text;            // stripped
  int64_t start_byte = 0;  // of its UTF-8 form inside the document's output, once encoded
};

bool is_space(char32_t c) {  // str.isspace(): bidirectional class WS / B / S or category Zs
  return (c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20) || c == 0x85 || c == 0xA0 || c == 0x1680 ||
         (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

Text stripped(const Text& s, size_t b, size_t e) {  // s[b:e].strip()
  while (b < e && is_space(s[b])) ++b;
  while (e > b && is_space(s[e - 1])) --e;
  return s.substr(b, e - b);
}

bool blank(const Text& s) {
  for (char32_t c : s)
    if (!is_space(c)) return false;
  return true;
}

Text decode(const char* p, size_t n) {  // UTF-8 -> code points; a malformed byte becomes U+FFFD
  Text out;
  out.reserve(n);
  size_t i = 0;
  while (i < n) {
    const unsigned char c = static_cast<unsigned char>(p[i]);
    if (c < 0x80) {
      out.push_back(c);
      ++i;
      continue;
    }
    const int extra = (c >> 5) == 0x6 ? 1 : (c >> 4) == 0xE ? 2 : (c >> 3) == 0x1E ? 3 : -1;
    if (extra < 0 || i + static_cast<size_t>(extra) >= n) {
      out.push_back(0xFFFD);
      ++i;
      continue;
    }
    char32_t cp = extra == 1 ? (c & 0x1F) : extra == 2 ? (c & 0x0F) : (c & 0x07);
    bool ok = true;
    for (int k = 1; k <= extra; ++k) {
      const unsigned char d = static_cast<unsigned char>(p[i + static_cast<size_t>(k)]);
      if ((d >> 6) != 0x2) {
        ok = false;
        break;
      }
      cp = (cp << 6) | (d & 0x3F);
    }
    if (!ok) {
      out.push_back(0xFFFD);
      ++i;
      continue;
    }
    out.push_back(cp);
    i += static_cast<size_t>(extra) + 1;
  }
  return out;
}

void encode(const Text& s, std::string* out) {
  for (char32_t c : s) {
    if (c < 0x80) {
      out->push_back(static_cast<char>(c));
    } else if (c < 0x800) {
      out->push_back(static_cast<char>(0xC0 | (c >> 6)));
      out->push_back(static_cast<char>(0x80 | (c & 0x3F)));
    } else if (c < 0x10000) {
      out->push_back(static_cast<char>(0xE0 | (c >> 12)));
      out->push_back(static_cast<char>(0x80 | ((c >> 6) & 0x3F)));
      out->push_back(static_cast<char>(0x80 | (c & 0x3F)));
    } else {
      out->push_back(static_cast<char>(0xF0 | (c >> 18)));
      out->push_back(static_cast<char>(0x80 | ((c >> 12) & 0x3F)));
      out->push_back(static_cast<char>(0x80 | ((c >> 6) & 0x3F)));
      out->push_back(static_cast<char>(0x80 | (c & 0x3F)));
    }
  }
}

struct Chunker {
  int64_t size, overlap;
  std::vector<Piece>* out;
  bool stuck = false;  // reached the size windows with chunk_overlap >= chunk_size

  void emit(const Text& s, size_t b, size_t e, int64_t start, int64_t end) {
    Text t = stripped(s, b, e);
    if (!t.empty()) out->push_back(Piece{start, end, std::move(t)});
  }

  // chunking.py:170-191: windows of `size`, stepping size - overlap; an all-blank window is skipped
  void by_size(const Text& s, int64_t base) {
    const int64_t n = static_cast<int64_t>(s.size());
    if (overlap >= size) {  // :187 pos += chunk_size - chunk_overlap never advances: the reference hangs here
      stuck = true;
      return;
    }
    for (int64_t pos = 0; pos < n; pos += size - overlap) {
      const int64_t end = std::min(pos + size, n);
      emit(s, static_cast<size_t>(pos), static_cast<size_t>(end), base + pos, base + end);
    }
  }

  // chunking.py:68-168. `level` indexes the separator list; the reference passes the list's tail.
  void recursive(const Text& s, int level, int64_t base) {
    static const Text kSeps[] = {U"\n\n", U"\n", U". ", U"? ", U"! ", U"; ", U", ", U" "};
    constexpr int kNumSeps = 8;
    const int64_t n = static_cast<int64_t>(s.size());
    if (n == 0) return;
    if (n <= size) {  // :79-90 fits: one chunk, offsets of the unstripped text
      emit(s, 0, s.size(), base, base + n);
      return;
    }
    int lv = level;  // :93-97 the first separator that occurs; the final "" always "occurs"
    while (lv < kNumSeps && s.find(kSeps[lv]) == Text::npos) ++lv;
    if (lv >= kNumSeps) {  // :99-102
      by_size(s, base);
      return;
    }
    const Text& sep = kSeps[lv];
    // :105-107; parts are walked as (begin, length) over s instead of materialised
    Text cur;
    int64_t cur_start = base;
    size_t from = 0;
    for (;;) {
      const size_t hit = s.find(sep, from);
      const bool last = hit == Text::npos;
      const size_t pe = last ? s.size() : hit + sep.size();  // end of part_with_sep (:110-111)
      const int64_t plen = static_cast<int64_t>(pe - from);
      if (static_cast<int64_t>(cur.size()) + plen <= size) {  // :114-115
        cur.append(s, from, pe - from);
      } else {
        if (!blank(cur)) emit(cur, 0, cur.size(), cur_start, cur_start + static_cast<int64_t>(cur.size()));  // :117-126
        if (overlap > 0 && !cur.empty()) {
          // :129-135. The reference recomputes current_start from the ALREADY reassigned chunk, so the
          // three lengths cancel and the start of an overlapped chunk stays where the previous one began.
          const size_t keep = std::min<size_t>(cur.size(), static_cast<size_t>(overlap));
          Text next = cur.substr(cur.size() - keep);
          next.append(s, from, pe - from);
          cur.swap(next);
        } else {  // :136-141
          cur.assign(s, from, pe - from);
          cur_start = base + static_cast<int64_t>(from);  // sum of len(part) + len(sep) over the parts before
        }
        if (plen > size) {  // :144-155 a single part that is too long goes down one separator level
          recursive(s.substr(from, pe - from), lv + 1, cur_start);
          cur.clear();
        }
      }
      if (last) break;
      from = pe;
    }
    if (!blank(cur)) emit(cur, 0, cur.size(), cur_start, cur_start + static_cast<int64_t>(cur.size()));  // :158-168
  }

  // chunking.py:193-239
  void sentences(const Text& s) {
    Text cur;
    int64_t cur_start = 0, pos = 0;
    auto flush = [&] {
      if (!cur.empty()) out->push_back(Piece{cur_start, cur_start + static_cast<int64_t>(cur.size()), cur});
    };
    const size_t n = s.size();
    size_t b = 0;
    while (b <= n) {
      // next split point of re.split(r"(?<=[.!?])\s+"): a white-space run right after . ! or ?
      size_t e = b, next = n + 1;
      for (size_t i = std::max<size_t>(b, 1); i < n; ++i) {
        if (is_space(s[i]) && (s[i - 1] == U'.' || s[i - 1] == U'!' || s[i - 1] == U'?')) {
          e = i;
          size_t j = i;
          while (j < n && is_space(s[j])) ++j;
          next = j;
          break;
        }
      }
      if (next == n + 1) e = n;
      const Text sent = stripped(s, b, e);  // :206
      if (!sent.empty()) {
        if (static_cast<int64_t>(cur.size() + sent.size()) + 1 <= size) {  // :210-215
          if (!cur.empty()) {
            cur.push_back(U' ');
            cur += sent;
          } else {
            cur = sent;
            cur_start = pos;
          }
        } else {  // :216-227
          flush();
          cur = sent;
          cur_start = pos;
        }
        const size_t at = s.find(sent, static_cast<size_t>(pos));  // :229
        pos = (at == Text::npos ? -1 : static_cast<int64_t>(at)) + static_cast<int64_t>(sent.size());
      }
      if (next == n + 1) break;
      b = next;
    }
    flush();  // :231-239
  }
};

}  // namespace

struct vr_chunks {
  std::vector<int64_t> doc_off, span, text_off;
  std::string text;
};

extern "C" {

int vr_chunk_texts(const char* const* texts, const int64_t* text_lens, int64_t n_texts, int32_t chunk_size,
                   int32_t chunk_overlap, int32_t strategy, vr_chunks** out) {
  VR_CHECK(out && n_texts >= 0 && (n_texts == 0 || (texts && text_lens)), "bad arguments");
  VR_CHECK(chunk_size > 0 && chunk_overlap >= 0, "chunk_size %d / chunk_overlap %d out of range", chunk_size, chunk_overlap);
  std::vector<std::vector<Piece>> per_doc(static_cast<size_t>(n_texts));
  std::vector<std::string> per_doc_text(static_cast<size_t>(n_texts));
  std::atomic<bool> stuck{false};
  vr::parallel_for(n_texts, 1, [&](int64_t d) {
    const Text s = decode(texts[d], static_cast<size_t>(text_lens[d]));
    if (s.empty() || blank(s)) return;  // chunking.py:35-36
    Chunker c{chunk_size, chunk_overlap, &per_doc[static_cast<size_t>(d)]};
    if (strategy == VR_CHUNK_SENTENCE) c.sentences(s);
    else if (strategy == VR_CHUNK_FIXED) c.by_size(s, 0);
    else c.recursive(s, 0, 0);  // "recursive" and every unknown name (chunking.py:38-45)
    if (c.stuck) stuck = true;
    // back to UTF-8 here, on the document's thread: the serial part below only copies bytes
    std::string& bytes = per_doc_text[static_cast<size_t>(d)];
    for (Piece& p : per_doc[static_cast<size_t>(d)]) {
      const size_t before = bytes.size();
      encode(p.text, &bytes);
      p.start_byte = static_cast<int64_t>(before);
      p.text.clear();
    }
  });
  // An overlap that reaches the chunk size is legal as long as no text falls through to the size
  // windows (the default overlap 50 with a small chunk_size, say); where one does, the reference's
  // loop never advances (chunking.py:187) and hangs, and this call fails instead.
  VR_CHECK(!stuck, "chunk_overlap %d >= chunk_size %d: the size windows cannot advance", chunk_overlap, chunk_size);
  vr_chunks* r = new vr_chunks();
  r->doc_off.assign(1, 0);
  r->text_off.assign(1, 0);
  size_t total_bytes = 0, total_chunks = 0;
  for (int64_t d = 0; d < n_texts; ++d) {
    total_bytes += per_doc_text[static_cast<size_t>(d)].size();
    total_chunks += per_doc[static_cast<size_t>(d)].size();
  }
  r->text.reserve(total_bytes);
  r->span.reserve(2 * total_chunks);
  r->text_off.reserve(total_chunks + 1);
  for (int64_t d = 0; d < n_texts; ++d) {
    const int64_t base = static_cast<int64_t>(r->text.size());
    const std::string& bytes = per_doc_text[static_cast<size_t>(d)];
    const std::vector<Piece>& doc = per_doc[static_cast<size_t>(d)];
    for (size_t i = 0; i < doc.size(); ++i) {
      r->span.push_back(doc[i].start);
      r->span.push_back(doc[i].end);
      const int64_t end = i + 1 < doc.size() ? doc[i + 1].start_byte : static_cast<int64_t>(bytes.size());
      r->text_off.push_back(base + end);
    }
    r->text += bytes;
    r->doc_off.push_back(static_cast<int64_t>(r->span.size() / 2));
  }
  *out = r;
  return 0;
}

int vr_chunks_view(const vr_chunks* c, int64_t* n_chunks, const int64_t** doc_off, const int64_t** span,
                   const int64_t** text_off, const char** text) {
  VR_CHECK(c && n_chunks && doc_off && span && text_off && text, "bad arguments");
  *n_chunks = static_cast<int64_t>(c->span.size() / 2);
  *doc_off = c->doc_off.data();
  *span = c->span.data();
  *text_off = c->text_off.data();
  *text = c->text.data();
  return 0;
}

void vr_chunks_free(vr_chunks* c) { delete c; }

}  // extern "C"
