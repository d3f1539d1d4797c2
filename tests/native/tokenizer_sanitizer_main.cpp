// AddressSanitizer / UBSan driver for rag_fin_amd/csrc/tokenizer.cpp (a
// sanitizer build of the host-side code; GPU sanitizers are not available on the pool).
// Built and run by tests/test_tokenizer_sanitizer.py on the CPU:
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all tokenizer.cpp this.cpp
// Input file (little endian): int64 vocab_bytes, vocab, int64 n_punct, int32 punct[n_punct],
// int64 n, int64 offsets[n + 1], text bytes.
// Output file: int32 ids[n][max_len], int32 lens[n].
// Besides the batch it drives the argument-check paths with null / empty / degenerate arguments.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ragfin.h"

// the library's error sink lives in index.hip (a HIP translation unit): stub for the host-only build
static char g_err[512];
void rf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::vector<char> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s <input> <max_len> <threads> <output>\n", argv[0]);
    return 2;
  }
  const std::vector<char> in = slurp(argv[1]);
  const int max_len = atoi(argv[2]), threads = atoi(argv[3]);
  const char* p = in.data();
  int64_t vb, n, np_;
  memcpy(&vb, p, 8); p += 8;
  const char* vocab = p; p += vb;
  memcpy(&np_, p, 8); p += 8;
  std::vector<int32_t> punct((size_t)np_);
  memcpy(punct.data(), p, (size_t)np_ * 4); p += np_ * 4;
  memcpy(&n, p, 8); p += 8;
  std::vector<int64_t> off((size_t)n + 1);
  memcpy(off.data(), p, (size_t)(n + 1) * 8); p += (n + 1) * 8;
  const char* text = p;

  rf_tokenizer_t* t = nullptr;
  // argument checks first: none of these may touch memory
  if (rf_tokenizer_create(nullptr, vocab, (size_t)vb, 1, 100) == RF_OK) return 3;
  if (rf_tokenizer_create(&t, nullptr, 0, 1, 100) == RF_OK) return 3;
  if (rf_tokenizer_create(&t, vocab, 0, 1, 100) == RF_OK) return 3;
  if (rf_tokenizer_create(&t, "no-specials\nhere\n", 17, 1, 100) == RF_OK) return 3;
  if (rf_tokenizer_create(&t, vocab, (size_t)vb, 1, 100) != RF_OK) {
    fprintf(stderr, "create failed: %s\n", g_err);
    return 4;
  }
  int32_t sp[5];
  if (rf_tokenizer_special_ids(t, sp) != RF_OK) return 4;
  if (rf_tokenizer_set_punctuation(t, punct.data(), (int)np_) != RF_OK) return 4;
  std::vector<int32_t> ids((size_t)n * max_len), lens((size_t)n);
  if (rf_tokenize_batch(t, text, off.data(), (int)n, 1, ids.data(), lens.data(), threads) == RF_OK) return 3;   // max_len < 2
  if (rf_tokenize_batch(nullptr, text, off.data(), (int)n, max_len, ids.data(), lens.data(), threads) == RF_OK) return 3;
  if (rf_tokenize_batch(t, text, off.data(), 0, max_len, ids.data(), lens.data(), threads) != RF_OK) return 4;   // empty batch
  if (rf_tokenize_batch(t, text, off.data(), (int)n, max_len, ids.data(), lens.data(), threads) != RF_OK) {
    fprintf(stderr, "tokenize failed: %s\n", g_err);
    return 4;
  }
  for (int64_t i = 0; i < n; ++i)
    if (lens[(size_t)i] < 2 || lens[(size_t)i] > max_len) return 5;
  FILE* f = fopen(argv[4], "wb");
  if (!f) return 2;
  fwrite(ids.data(), 4, ids.size(), f);
  fwrite(lens.data(), 4, lens.size(), f);
  fclose(f);
  rf_tokenizer_destroy(t);
  rf_tokenizer_destroy(nullptr);
  return 0;
}
