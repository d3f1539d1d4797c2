// Native BERT WordPiece tokenizer (host code, multi-threaded): the text -> token-id stage in
// front of rf_encode.  Stands in for the tokenizer that
// SentenceTransformer('all-MiniLM-L6-v2') loads by name (vector_rag_mcp/main.py:41,50;
// "chunking_storing (1).py":8,380): BERT "basic" tokenisation + greedy longest-match
// WordPiece with the "##" continuation prefix, [CLS] ... [SEP], truncation to max_len.
//
// Division of labour with rag_fin_amd/tokenizer.py (the Python restatement that is checked
// against transformers.BertTokenizer and is the parity reference for this file):
//   * ASCII text is handled here end to end: control characters dropped, " \t\n\r" are the
//     whitespace, A-Z lower-cased, the four ASCII punctuation ranges split off;
//   * text with non-ASCII characters is PRE-NORMALISED by the Python side with C-speed string
//     primitives (clean, NFC, lower, NFD, strip Mn, CJK and non-ASCII punctuation padded with
//     spaces), after which every remaining non-ASCII code point is a plain word character and
//     the rules above are all that is left to apply.  Unicode tables therefore stay in
//     Python's unicodedata; this file only needs to walk UTF-8 sequences.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ragfin.h"

void rf_set_error(const char* fmt, ...);

struct rf_tokenizer {
  std::unordered_map<std::string, int32_t> first;   // tokens that start a word
  std::unordered_map<std::string, int32_t> cont;    // "##x" continuation tokens, keyed by x
  int32_t unk = -1, cls = -1, sep = -1, pad = -1, mask = -1;
  bool lower = true;
  int max_chars = 100;
  std::vector<int32_t> punct;   // sorted non-ASCII code points to split off like ASCII punctuation
};

static inline bool ascii_punct(unsigned char c) {
  return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}
static inline bool ascii_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
// ASCII control characters other than \t \n \r are dropped (category Cc), as is DEL
static inline bool ascii_dropped(unsigned char c) { return (c < 0x20 && !ascii_space(c)) || c == 0x7f; }
static inline int utf8_len(unsigned char lead) {
  return lead < 0x80 ? 1 : (lead >> 5) == 0x6 ? 2 : (lead >> 4) == 0xe ? 3 : (lead >> 3) == 0x1e ? 4 : 1;
}

extern "C" int rf_tokenizer_create(rf_tokenizer_t** out, const char* vocab_utf8, size_t vocab_bytes,
                                   int do_lower_case, int max_chars_per_word) {
  if (!out || !vocab_utf8 || vocab_bytes == 0) {
    rf_set_error("rf_tokenizer_create: null / empty vocabulary");
    return RF_ERR_INVALID;
  }
  *out = nullptr;
  rf_tokenizer* t = new (std::nothrow) rf_tokenizer();
  if (!t) {
    rf_set_error("out of host memory");
    return RF_ERR_INVALID;
  }
  t->lower = do_lower_case != 0;
  t->max_chars = max_chars_per_word > 0 ? std::min(max_chars_per_word, 126) : 100;
  int32_t id = 0;
  size_t p = 0;
  while (p < vocab_bytes) {
    size_t e = p;
    while (e < vocab_bytes && vocab_utf8[e] != '\n') ++e;
    size_t len = e - p;
    if (len > 0 && vocab_utf8[p + len - 1] == '\r') --len;   // tolerate CRLF files
    std::string tok(vocab_utf8 + p, len);
    // a duplicate line keeps the LAST id, like the Python dict comprehension
    if (tok.size() > 2 && tok[0] == '#' && tok[1] == '#') t->cont[tok.substr(2)] = id;
    t->first[tok] = id;   // "##x" is also reachable as a literal word (e.g. the text "##x" after punctuation splitting never is, but a special-token lookup may be)
    ++id;
    p = e + 1;
  }
  auto get = [&](const char* s) {
    auto it = t->first.find(s);
    return it == t->first.end() ? -1 : it->second;
  };
  t->unk = get("[UNK]");
  t->cls = get("[CLS]");
  t->sep = get("[SEP]");
  t->pad = get("[PAD]");
  t->mask = get("[MASK]");
  if (t->unk < 0 || t->cls < 0 || t->sep < 0 || t->pad < 0) {
    delete t;
    rf_set_error("rf_tokenizer_create: vocabulary lacks [UNK] / [CLS] / [SEP] / [PAD]");
    return RF_ERR_INVALID;
  }
  *out = t;
  return RF_OK;
}

extern "C" int rf_tokenizer_destroy(rf_tokenizer_t* t) {
  delete t;
  return RF_OK;
}

extern "C" int rf_tokenizer_set_punctuation(rf_tokenizer_t* t, const int32_t* cps, int n) {
  if (!t || n < 0 || (n > 0 && !cps)) {
    rf_set_error("rf_tokenizer_set_punctuation: bad argument");
    return RF_ERR_INVALID;
  }
  t->punct.assign(cps, cps + n);
  std::sort(t->punct.begin(), t->punct.end());
  return RF_OK;
}

extern "C" int rf_tokenizer_special_ids(const rf_tokenizer_t* t, int32_t* ids5) {
  if (!t || !ids5) return RF_ERR_INVALID;
  ids5[0] = t->unk;
  ids5[1] = t->cls;
  ids5[2] = t->sep;
  ids5[3] = t->pad;
  ids5[4] = t->mask;
  return RF_OK;
}

namespace {
struct Scratch {
  std::string word;   // cleaned, lower-cased bytes of the current whitespace-delimited word
  std::string key;
  std::vector<int32_t> ids;
};

// greedy longest-match WordPiece of one punctuation-free piece [s, s+n) (UTF-8)
void wordpiece(const rf_tokenizer& t, const char* s, size_t n, Scratch& sc) {
  // code-point boundaries
  size_t nchar = 0;
  size_t bounds[128];
  for (size_t i = 0; i < n;) {
    if (nchar < 127) bounds[nchar] = i;
    ++nchar;
    i += (size_t)utf8_len((unsigned char)s[i]);
  }
  if (nchar > (size_t)t.max_chars || nchar > 126) {
    sc.ids.push_back(t.unk);
    return;
  }
  bounds[nchar] = n;
  const size_t mark = sc.ids.size();
  size_t start = 0;
  while (start < nchar) {
    size_t end = nchar;
    int32_t cur = -1;
    while (start < end) {
      sc.key.assign(s + bounds[start], bounds[end] - bounds[start]);
      const auto& map = start > 0 ? t.cont : t.first;
      auto it = map.find(sc.key);
      if (it != map.end()) {
        cur = it->second;
        break;
      }
      --end;
    }
    if (cur < 0) {   // the whole word becomes [UNK]
      sc.ids.resize(mark);
      sc.ids.push_back(t.unk);
      return;
    }
    sc.ids.push_back(cur);
    start = end;   // (the word is always finished: a later failure turns ALL its pieces into one [UNK])
  }
}

bool is_special(const std::string& w) {
  return w == "[UNK]" || w == "[CLS]" || w == "[SEP]" || w == "[PAD]" || w == "[MASK]";
}

// one text -> row of ids (with [CLS] / [SEP]); returns the row length
int encode_one(const rf_tokenizer& t, const char* s, size_t n, int max_len, int32_t* row, Scratch& sc) {
  sc.ids.clear();
  const size_t limit = max_len > 2 ? (size_t)(max_len - 2) : 0;
  size_t i = 0;
  while (i < n && sc.ids.size() < limit) {
    // next whitespace-delimited word, control characters dropped
    // (dropped characters do not separate words; before a word they are simply skipped)
    while (i < n && (ascii_space((unsigned char)s[i]) || ascii_dropped((unsigned char)s[i]))) ++i;
    sc.word.clear();
    while (i < n && !ascii_space((unsigned char)s[i])) {
      const unsigned char c = (unsigned char)s[i];
      if (!ascii_dropped(c)) sc.word.push_back((char)c);
      ++i;
    }
    if (sc.word.empty()) continue;
    if (is_special(sc.word)) {
      auto it = t.first.find(sc.word);
      sc.ids.push_back(it == t.first.end() ? t.unk : it->second);
      continue;
    }
    if (t.lower)
      for (char& ch : sc.word)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
    // split off ASCII punctuation; everything between is one WordPiece input
    const char* w = sc.word.data();
    const size_t wn = sc.word.size();
    size_t a = 0;
    // punctuation = the four ASCII ranges, plus the non-ASCII code points the caller registered
    // (rf_tokenizer_set_punctuation); returns the byte length of the punctuation character at p, or 0
    auto punct_len = [&](size_t p) -> size_t {
      const unsigned char c0 = (unsigned char)w[p];
      if (c0 < 0x80) return ascii_punct(c0) ? 1 : 0;
      if (t.punct.empty()) return 0;
      const int len = utf8_len(c0);
      if (p + (size_t)len > wn) return 0;
      int32_t cp = len == 2 ? (c0 & 0x1f) : len == 3 ? (c0 & 0x0f) : (c0 & 0x07);
      for (int q = 1; q < len; ++q) cp = (cp << 6) | ((unsigned char)w[p + q] & 0x3f);
      return std::binary_search(t.punct.begin(), t.punct.end(), cp) ? (size_t)len : 0;
    };
    while (a < wn && sc.ids.size() < limit) {
      const size_t pl = punct_len(a);
      if (pl) {
        wordpiece(t, w + a, pl, sc);
        a += pl;
        continue;
      }
      size_t b = a;
      while (b < wn && !punct_len(b)) b += (size_t)utf8_len((unsigned char)w[b]);
      if (b > wn) b = wn;
      wordpiece(t, w + a, b - a, sc);
      a = b;
    }
  }
  const size_t keep = std::min(sc.ids.size(), limit);   // max_len >= 2 (checked by the caller)
  int len = 0;
  row[len++] = t.cls;
  for (size_t j = 0; j < keep; ++j) row[len++] = sc.ids[j];
  row[len++] = t.sep;
  return len;
}
}  // namespace

// Character offsets (code points, what len(str) counts) -> byte offsets into the texts' UTF-8 bytes laid back to
// back: the host then hands over ONE encoded blob of a whole batch and the per-text character counts instead of
// encoding and measuring every text on its own.  One pass; a code point starts at every byte that is not 10xxxxxx.
extern "C" int rf_utf8_offsets(const char* text_bytes, int64_t n_bytes, const int64_t* char_offsets, int n,
                               int64_t* byte_offsets) {
  if (!text_bytes || !char_offsets || !byte_offsets || n < 0 || n_bytes < 0) return -1;
  int64_t chars = 0, pos = 0;
  int j = 0;
  while (j <= n && char_offsets[j] == 0) byte_offsets[j++] = 0;
  for (; pos < n_bytes && j <= n; ++pos) {
    if (((unsigned char)text_bytes[pos] & 0xC0) != 0x80) {   // a code point starts here
      while (j <= n && char_offsets[j] == chars) byte_offsets[j++] = pos;
      ++chars;
    }
  }
  while (j <= n && char_offsets[j] == chars) byte_offsets[j++] = n_bytes;
  return j == n + 1 ? 0 : -1;   // the counts do not add up to the blob (or are not ascending)
}

extern "C" int rf_tokenize_batch(const rf_tokenizer_t* t, const char* text_bytes, const int64_t* offsets, int n,
                                 int max_len, int32_t* ids_out, int32_t* lens_out, int n_threads) {
  if (!t || !offsets || !ids_out || !lens_out || n < 0 || max_len < 2 || (n > 0 && !text_bytes && offsets[n] > 0)) {
    rf_set_error("rf_tokenize_batch: bad argument (max_len must be >= 2)");
    return RF_ERR_INVALID;
  }
  for (int i = 0; i < n; ++i)
    if (offsets[i + 1] < offsets[i]) {
      rf_set_error("rf_tokenize_batch: offsets not monotone at %d", i);
      return RF_ERR_INVALID;
    }
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 64) nt = 64;
  if (nt > (n + 63) / 64) nt = std::max(1, (n + 63) / 64);   // at least ~64 texts per thread
  std::atomic<int> next{0};
  auto work = [&]() {
    Scratch sc;
    for (;;) {
      const int i0 = next.fetch_add(16);
      if (i0 >= n) break;
      const int i1 = std::min(n, i0 + 16);
      for (int i = i0; i < i1; ++i) {
        int32_t* row = ids_out + (size_t)i * max_len;
        const int len = encode_one(*t, text_bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), max_len, row, sc);
        for (int j = len; j < max_len; ++j) row[j] = t->pad;
        lens_out[i] = len;
      }
    }
  };
  if (nt == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    for (int k = 0; k < nt; ++k) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  return RF_OK;
}
