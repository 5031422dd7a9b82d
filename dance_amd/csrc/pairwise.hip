// Dense pairwise distance matrix — replaces the numba kernel
// dance/utils/matrix.py:164-180 used by SpaGCNGraph / SpaGCNGraph2D (spatial_graph.py:60,75).
//
// out[i][j] = dist(x_i, x_j) for all i, j (no symmetry shortcut, like the reference).
//   euclidean (matrix.py:100-105): each term (x_it - x_jt) is an f32 subtract and an f32 square;
//     the running sum starts as the Python int 0, which numba unifies with f32 to float64, so
//     the sum and the sqrt are double and only the return value rounds to f32.  Same here:
//     terms in f32, accumulation in index order in f64, one rounding at the end.
//   pearson (matrix.py:108-116): 1 - cov/sqrt(var_a var_b); evaluated in f64 from centred rows
//     (the reference's f32 intermediate roundings are below its own test tolerance, np.allclose).
//   spearman (matrix.py:119-157): pearson on mean-ranked rows; dh_rank_rows_f32 ranks the rows.
//
// The output is N^2 floats, so for the reference's d = 2..3 this kernel is HBM-write-bound:
// 64x64 output tile per block, each lane owns a 4x4 patch and writes 16-B rows.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 64;    // tile edge
constexpr int DC = 32;   // feature chunk staged in LDS

template <int METRIC>
__global__ __launch_bounds__(256) void pairwise_kernel(int64_t n, int64_t d, const float* __restrict__ X,
                                                       int64_t ldx, float* __restrict__ out, int64_t ldo) {
  __shared__ float xi[T][DC + 1];
  __shared__ float xj[T][DC + 1];
  __shared__ double mi[T], mj[T], vi[T], vj[T];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * T, j0 = (int64_t)blockIdx.x * T;
  const int ty = tid / 16, tx = tid % 16;  // lane owns rows i0 + ty*4 + a, cols j0 + tx*4 + b

  if constexpr (METRIC == DH_METRIC_PEARSON) {
    // row means and centred sums of squares of the 2 x 64 rows this block touches
    if (tid < 2 * T) {
      const bool is_j = tid >= T;
      const int r = tid % T;
      const int64_t row = (is_j ? j0 : i0) + r;
      double m = 0.0, v = 0.0;
      if (row < n) {
        const float* p = X + row * ldx;
        for (int64_t t = 0; t < d; ++t) m += (double)p[t];
        m /= (double)d;
        for (int64_t t = 0; t < d; ++t) { const double c = (double)p[t] - m; v += c * c; }
      }
      (is_j ? mj : mi)[r] = m;
      (is_j ? vj : vi)[r] = v;
    }
    __syncthreads();
  }

  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

  for (int64_t t0 = 0; t0 < d; t0 += DC) {
    for (int idx = tid; idx < T * DC; idx += 256) {
      const int r = idx / DC, c = idx % DC;
      const int64_t t = t0 + c;
      xi[r][c] = (i0 + r < n && t < d) ? X[(i0 + r) * ldx + t] : 0.f;
      xj[r][c] = (j0 + r < n && t < d) ? X[(j0 + r) * ldx + t] : 0.f;
    }
    __syncthreads();
    const int lim = (int)min((int64_t)DC, d - t0);
    for (int c = 0; c < lim; ++c) {
      float a4[4], b4[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) a4[a] = xi[ty * 4 + a][c];
#pragma unroll
      for (int b = 0; b < 4; ++b) b4[b] = xj[tx * 4 + b][c];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if constexpr (METRIC == DH_METRIC_EUCLIDEAN) {
            const float diff = __fsub_rn(a4[a], b4[b]);
            acc[a][b] += (double)__fmul_rn(diff, diff);
          } else {
            acc[a][b] += ((double)a4[a] - mi[ty * 4 + a]) * ((double)b4[b] - mj[tx * 4 + b]);
          }
        }
    }
    __syncthreads();
  }

#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int64_t i = i0 + ty * 4 + a;
    if (i >= n) continue;
    float r[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if constexpr (METRIC == DH_METRIC_EUCLIDEAN) r[b] = (float)sqrt(acc[a][b]);
      else r[b] = (float)(1.0 - acc[a][b] / sqrt(vi[ty * 4 + a] * vj[tx * 4 + b]));
    }
    const int64_t j = j0 + tx * 4;
    float* o = out + i * ldo + j;
    if (j + 3 < n && (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
      f32x4 v = {r[0], r[1], r[2], r[3]};
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o));
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (j + b < n) o[b] = r[b];
    }
  }
}

// Mean ranks with ties (scipy rankdata 'average', matrix.py:119-140): rank = #less + (#equal + 1) / 2.
__global__ __launch_bounds__(256) void rank_rows_kernel(int64_t n, int64_t d, const float* __restrict__ X,
                                                        int64_t ldx, float* __restrict__ out, int64_t ldo) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / d, c = i % d;
    const float* p = X + r * ldx;
    const float v = p[c];
    int less = 0, equal = 0;
    for (int64_t t = 0; t < d; ++t) {
      less += p[t] < v;
      equal += p[t] == v;
    }
    out[r * ldo + c] = (float)less + 0.5f * (float)(equal + 1);
  }
}

}  // namespace

extern "C" int dh_pairwise_distance_f32(int64_t n, int64_t d, const float* X, int64_t ldx, float* out,
                                        int64_t ldo, int metric, dh_stream_t stream) {
  if (n < 0 || d < 0) return dh::fail(DH_ERR_INVALID, "dh_pairwise_distance_f32: negative size");
  if (n == 0) return DH_OK;
  if (!X || !out) return dh::fail(DH_ERR_INVALID, "dh_pairwise_distance_f32: null pointer");
  if (ldx < d || ldo < n) return dh::fail(DH_ERR_INVALID, "dh_pairwise_distance_f32: leading dimension too small");
  if (n > (int64_t)65535 * T) return dh::fail(DH_ERR_INVALID, "dh_pairwise_distance_f32: n too large for a dense matrix");
  hipStream_t st = dh::as_stream(stream);
  dim3 grid((unsigned)dh::ceil_div(n, T), (unsigned)dh::ceil_div(n, T));
  switch (metric) {
    case DH_METRIC_EUCLIDEAN:
      hipLaunchKernelGGL(pairwise_kernel<DH_METRIC_EUCLIDEAN>, grid, dim3(256), 0, st, n, d, X, ldx, out, ldo);
      break;
    case DH_METRIC_PEARSON:
    case DH_METRIC_SPEARMAN:  // caller passes mean-ranked rows (dh_rank_rows_f32)
      hipLaunchKernelGGL(pairwise_kernel<DH_METRIC_PEARSON>, grid, dim3(256), 0, st, n, d, X, ldx, out, ldo);
      break;
    default:
      return dh::fail(DH_ERR_INVALID, "dh_pairwise_distance_f32: unknown metric %d", metric);
  }
  return dh::check_launch("dh_pairwise_distance_f32");
}

extern "C" int dh_rank_rows_f32(int64_t n, int64_t d, const float* X, int64_t ldx, float* out, int64_t ldo,
                                dh_stream_t stream) {
  if (n < 0 || d < 0) return dh::fail(DH_ERR_INVALID, "dh_rank_rows_f32: negative size");
  if (n == 0 || d == 0) return DH_OK;
  if (!X || !out) return dh::fail(DH_ERR_INVALID, "dh_rank_rows_f32: null pointer");
  if (ldx < d || ldo < d) return dh::fail(DH_ERR_INVALID, "dh_rank_rows_f32: leading dimension too small");
  const unsigned grid = (unsigned)(dh::ceil_div(n * d, 256) < 8192 ? dh::ceil_div(n * d, 256) : 8192);
  hipLaunchKernelGGL(rank_rows_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), n, d, X, ldx, out, ldo);
  return dh::check_launch("dh_rank_rows_f32");
}
