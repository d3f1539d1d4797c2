/*
 * search_oracle.c -- plain-C twin of oracle/search.py.  TEST INFRASTRUCTURE ONLY
 * (see the header of oracle/search.py: who may call it, what it restates, and
 * why parity with the reference's Milvus backend is unpinned).
 *
 * Restates the contract of the reference's
 *   collection.search(q, "embedding", {"metric_type": "COSINE"}, top_k)
 * (vector_rag_mcp/main.py:51-57): every row scored, best `k` by descending score.
 *
 * Ranking contract (oracle/search.py): eight interleaved float64 fma chains
 * (chain j takes d = j, j+8, ...), combined ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7));
 * order by (score desc, row asc).  No -ffast-math: the loop over rows may be
 * vectorised, the chains may not be re-associated.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* IEEE binary16 -> double, exact */
static double h2d(uint16_t h) {
  const uint32_t sign = (uint32_t)(h >> 15);
  const uint32_t exp = (h >> 10) & 0x1f;
  const uint32_t man = h & 0x3ff;
  double v;
  if (exp == 0) v = ldexp((double)man, -24);
  else if (exp == 31) v = man ? NAN : INFINITY;
  else v = ldexp((double)(man | 0x400), (int)exp - 25);
  return sign ? -v : v;
}

/* is (s1, r1) ranked strictly before (s2, r2)? */
static int before(double s1, int64_t r1, double s2, int64_t r2) {
  return (s1 > s2) || (s1 == s2 && r1 < r2);
}

/*
 * q: fp16 bits [B, D]; c: fp16 bits [N, D]; out_s: f64 [B, k]; out_i: i64 [B, k].
 * Slots past N get (-inf, -1).  Returns 0, or -1 on allocation failure.
 */
int oracle_search(const uint16_t* q, const uint16_t* c, int64_t B, int64_t N, int64_t D,
                  int64_t k, int64_t id_base, double* out_s, int64_t* out_i) {
  enum { TILE = 1024 };
  double* qd = (double*)malloc((size_t)D * sizeof(double));
  double* ct = (double*)malloc((size_t)D * TILE * sizeof(double)); /* [D][TILE] */
  double* acc = (double*)malloc((size_t)8 * TILE * sizeof(double)); /* [8][TILE] */
  if (!qd || !ct || !acc) { free(qd); free(ct); free(acc); return -1; }
  for (int64_t b = 0; b < B; ++b)
    for (int64_t j = 0; j < k; ++j) { out_s[b * k + j] = -INFINITY; out_i[b * k + j] = -1; }
  int64_t* fill = (int64_t*)calloc((size_t)B, sizeof(int64_t));
  if (!fill) { free(qd); free(ct); free(acc); return -1; }

  for (int64_t r0 = 0; r0 < N; r0 += TILE) {
    const int64_t m = (N - r0) < TILE ? (N - r0) : TILE;
    for (int64_t r = 0; r < m; ++r)
      for (int64_t d = 0; d < D; ++d) ct[d * TILE + r] = h2d(c[(r0 + r) * D + d]);
    for (int64_t b = 0; b < B; ++b) {
      for (int64_t d = 0; d < D; ++d) qd[d] = h2d(q[b * D + d]);
      for (int64_t r = 0; r < 8 * TILE; ++r) acc[r] = 0.0;
      for (int64_t d = 0; d < D; ++d) {
        const double qv = qd[d];
        const double* col = ct + d * TILE;
        double* pj = acc + (d & 7) * TILE; /* chain j = d mod 8 */
        for (int64_t r = 0; r < m; ++r) pj[r] = fma(qv, col[r], pj[r]);
      }
      for (int64_t r = 0; r < m; ++r)
        acc[r] = ((acc[r] + acc[TILE + r]) + (acc[2 * TILE + r] + acc[3 * TILE + r])) +
                 ((acc[4 * TILE + r] + acc[5 * TILE + r]) + (acc[6 * TILE + r] + acc[7 * TILE + r]));
      double* ls = out_s + b * k;
      int64_t* li = out_i + b * k;
      for (int64_t r = 0; r < m; ++r) {
        const double s = acc[r];
        const int64_t id = r0 + r + id_base;
        int64_t n = fill[b];
        if (n == k && !before(s, id, ls[k - 1], li[k - 1])) continue;
        int64_t pos = n < k ? n : k - 1;
        while (pos > 0 && before(s, id, ls[pos - 1], li[pos - 1])) {
          ls[pos] = ls[pos - 1];
          li[pos] = li[pos - 1];
          --pos;
        }
        ls[pos] = s;
        li[pos] = id;
        if (n < k) fill[b] = n + 1;
      }
    }
  }
  free(fill); free(qd); free(ct); free(acc);
  return 0;
}
