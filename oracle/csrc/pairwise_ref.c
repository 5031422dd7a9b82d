/* C restatement of the reference's only in-tree native kernel — TEST INFRASTRUCTURE (see oracle/__init__.py).
 *
 * dance/utils/matrix.py:100-105,164-180: numba `pairwise_distance(x, 0)` = for every (i, j):
 *     sum = 0                      (Python int: numba unifies it with the f32 terms to float64)
 *     for t: sum += (x[i][t] - x[j][t]) ** 2        (f32 subtract, f32 square)
 *     mat[i][j] = sqrt(sum)        (double sqrt, rounded to f32 by the f4 return type)
 * prange over i and j -> OpenMP here.  Built by oracle/Makefile into oracle/_build/libpairwise_ref.so; used by
 * tests (against the golden vectors and the numpy restatement) and as a multi-threaded CPU baseline for
 * scripts/bench_rows.py.  Never linked into the product.
 */
#include <math.h>
#include <stdint.h>

void oracle_pairwise_euclidean_f32(const float* x, int64_t n, int64_t d, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = 0; j < n; ++j) {
      double sum = 0.0;
      for (int64_t t = 0; t < d; ++t) {
        const float diff = x[i * d + t] - x[j * d + t];
        const float sq = diff * diff;
        sum += (double)sq;
      }
      out[i * n + j] = (float)sqrt(sum);
    }
  }
}

/* exact-kNN distance definition shared with dh_knn_bruteforce_f32: separately rounded f32 sub / mul / add in
 * feature order (compile with -ffp-contract=off). */
void oracle_sqdist_f32(const float* x, int64_t n, int64_t d, int64_t q_begin, int64_t q_end, float* out /* [q, n] */) {
#pragma omp parallel for schedule(static)
  for (int64_t q = q_begin; q < q_end; ++q) {
    for (int64_t j = 0; j < n; ++j) {
      float acc = 0.f;
      for (int64_t t = 0; t < d; ++t) {
        const float diff = x[q * d + t] - x[j * d + t];
        const float sq = diff * diff;
        acc = acc + sq;
      }
      out[(q - q_begin) * n + j] = acc;
    }
  }
}
