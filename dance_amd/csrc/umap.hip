// UMAP fuzzy-simplicial-set connectivities on a kNN list — the
// arithmetic sc.pp.neighbors(method="umap") runs after its kNN search
// (dance/transforms/graph/neighbor_graph.py:52-55 -> scanpy 1.10.1 -> umap-learn:
// smooth_knn_dist + compute_membership_strengths + the t-conorm symmetrisation
// W + W^T - W o W^T with set_op_mix_ratio = 1, local_connectivity = 1, bandwidth = 1).
// The third-party source is not in the reference tree; this restates its published algorithm
// (constants SMOOTH_K_TOLERANCE = 1e-5, MIN_K_DIST_SCALE = 1e-3, 64 bisection steps).
//
// Stages (each an exported entry point; the host only sequences them and sizes buffers):
//   dh_umap_membership_f32   rho_i, sigma_i (f64 bisection, stored f32) and w_ij per kNN slot
//   dh_knn_row_nnz           per row: number of slots with w != 0 (self / underflow dropped)
//   dh_knn_graph_to_csr      rows sorted by column -> CSR of W
//   dh_csr_union_count/fill  sorted merge of W and W^T rows: v = (a + b) - a*b in f32
#include "common.h"

namespace {

constexpr double kSmoothKTol = 1e-5;
constexpr double kMinKDistScale = 1e-3;
constexpr int kIters = 64;

// deterministic single-block sum of n*k floats in double -> ws[0] = mean
__global__ __launch_bounds__(1024) void mean_kernel(int64_t total, const float* __restrict__ v, double* __restrict__ ws) {
  __shared__ double part[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < total; i += 1024) s += (double)v[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[0] = total > 0 ? part[0] / (double)total : 0.0;
}

__global__ __launch_bounds__(256) void membership_kernel(int64_t n, int k, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ dist, float* __restrict__ out_w,
                                                         float* __restrict__ out_sigma, float* __restrict__ out_rho,
                                                         const double* __restrict__ mean_all) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* di = dist + i * k;
  const int32_t* ii = idx + i * k;
  const double target = log2((double)k);

  // rho: first strictly positive distance in list order (local_connectivity = 1)
  float rho = 0.f;
  for (int j = 0; j < k; ++j)
    if (di[j] > 0.f) { rho = di[j]; break; }

  double lo = 0.0, hi = __longlong_as_double(0x7ff0000000000000LL), mid = 1.0;
  for (int it = 0; it < kIters; ++it) {
    double psum = 0.0;
    for (int j = 1; j < k; ++j) {
      const float dd = __fsub_rn(di[j], rho);
      psum += (dd > 0.f) ? exp(-((double)dd / mid)) : 1.0;
    }
    if (fabs(psum - target) < kSmoothKTol) break;
    if (psum > target) {
      hi = mid;
      mid = (lo + hi) / 2.0;
    } else {
      lo = mid;
      if (isinf(hi)) mid *= 2.0; else mid = (lo + hi) / 2.0;
    }
  }
  float sigma = (float)mid;
  if (rho > 0.f) {
    double m = 0.0;
    for (int j = 0; j < k; ++j) m += (double)di[j];
    m /= (double)k;
    if ((double)sigma < kMinKDistScale * m) sigma = (float)(kMinKDistScale * m);
  } else {
    if ((double)sigma < kMinKDistScale * mean_all[0]) sigma = (float)(kMinKDistScale * mean_all[0]);
  }
  out_sigma[i] = sigma;
  out_rho[i] = rho;

  for (int j = 0; j < k; ++j) {
    float w;
    const int32_t c = ii[j];
    if (c < 0) w = 0.f;                 // missing neighbour slot
    else if ((int64_t)c == i) w = 0.f;  // self edge
    else {
      const float dd = __fsub_rn(di[j], rho);
      if (dd <= 0.f || sigma == 0.f) w = 1.f;
      else w = expf(-__fdiv_rn(dd, sigma));
    }
    out_w[i * k + j] = w;
  }
}

__global__ __launch_bounds__(256) void knn_row_nnz_kernel(int64_t n, int k, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ w, int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int c = 0;
  for (int j = 0; j < k; ++j) c += (idx[i * k + j] >= 0 && w[i * k + j] != 0.f);
  counts[i] = c;
}

// one thread per row: selection by rank among the row's kept entries (k is small: O(k^2) compares)
__global__ __launch_bounds__(256) void knn_to_csr_kernel(int64_t n, int k, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ w, const int32_t* __restrict__ rowptr,
                                                         int32_t* __restrict__ out_col, float* __restrict__ out_val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t* ii = idx + i * k;
  const float* wi = w + i * k;
  const int base = rowptr[i];
  for (int j = 0; j < k; ++j) {
    const int32_t c = ii[j];
    if (c < 0 || wi[j] == 0.f) continue;
    int rank = 0;
    for (int t = 0; t < k; ++t) {
      const int32_t ct = ii[t];
      if (ct < 0 || wi[t] == 0.f) continue;
      rank += (ct < c) || (ct == c && t < j);
    }
    out_col[base + rank] = c;
    out_val[base + rank] = wi[j];
  }
}

// |cols(A_i) U cols(B_i)| for sorted rows
__global__ __launch_bounds__(256) void union_count_kernel(int64_t n, const int32_t* __restrict__ rpa,
                                                          const int32_t* __restrict__ ca, const int32_t* __restrict__ rpb,
                                                          const int32_t* __restrict__ cb, int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int a = rpa[i], ae = rpa[i + 1], b = rpb[i], be = rpb[i + 1], c = 0;
  while (a < ae && b < be) {
    const int x = ca[a], y = cb[b];
    a += x <= y;
    b += y <= x;
    ++c;
  }
  counts[i] = c + (ae - a) + (be - b);
}

__global__ __launch_bounds__(256) void fuzzy_union_fill_kernel(int64_t n, const int32_t* __restrict__ rpa,
                                                               const int32_t* __restrict__ ca, const float* __restrict__ va,
                                                               const int32_t* __restrict__ rpb, const int32_t* __restrict__ cb,
                                                               const float* __restrict__ vb, const int32_t* __restrict__ rpo,
                                                               int32_t* __restrict__ co, float* __restrict__ vo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int a = rpa[i], ae = rpa[i + 1], b = rpb[i], be = rpb[i + 1], o = rpo[i];
  while (a < ae || b < be) {
    const int x = a < ae ? ca[a] : 0x7fffffff, y = b < be ? cb[b] : 0x7fffffff;
    float wa = 0.f, wb = 0.f;
    int c;
    if (x <= y) { c = x; wa = va[a++]; }
    else c = y;
    if (y <= x) wb = vb[b++];
    co[o] = c;
    vo[o] = __fsub_rn(__fadd_rn(wa, wb), __fmul_rn(wa, wb));  // (W + W^T) - W o W^T, f32 like scipy
    ++o;
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)dh::ceil_div(n, 256); }

}  // namespace

extern "C" int dh_umap_membership_f32(int64_t n, int k, const int32_t* knn_idx, const float* knn_dist, float* out_w,
                                      float* out_sigma, float* out_rho, void* workspace, size_t workspace_bytes,
                                      dh_stream_t stream) {
  if (n < 0 || k < 0) return dh::fail(DH_ERR_INVALID, "dh_umap_membership_f32: negative size");
  if (n == 0 || k == 0) return DH_OK;
  if (!knn_idx || !knn_dist || !out_w || !out_sigma || !out_rho)
    return dh::fail(DH_ERR_INVALID, "dh_umap_membership_f32: null pointer");
  if (!workspace || workspace_bytes < sizeof(double))
    return dh::fail(DH_ERR_WORKSPACE, "dh_umap_membership_f32: workspace must hold 8 bytes");
  hipStream_t st = dh::as_stream(stream);
  double* ws = static_cast<double*>(workspace);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, st, n * (int64_t)k, knn_dist, ws);
  hipLaunchKernelGGL(membership_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, k, knn_idx, knn_dist, out_w,
                     out_sigma, out_rho, ws);
  return dh::check_launch("dh_umap_membership_f32");
}

extern "C" int dh_knn_row_nnz(int64_t n, int k, const int32_t* knn_idx, const float* w, int32_t* out_counts,
                              dh_stream_t stream) {
  if (n < 0 || k < 0) return dh::fail(DH_ERR_INVALID, "dh_knn_row_nnz: negative size");
  if (n == 0) return DH_OK;
  if (!knn_idx || !w || !out_counts) return dh::fail(DH_ERR_INVALID, "dh_knn_row_nnz: null pointer");
  hipLaunchKernelGGL(knn_row_nnz_kernel, dim3(blocks_for(n)), dim3(256), 0, dh::as_stream(stream), n, k, knn_idx, w, out_counts);
  return dh::check_launch("dh_knn_row_nnz");
}

extern "C" int dh_knn_graph_to_csr(int64_t n, int k, const int32_t* knn_idx, const float* w, const int32_t* rowptr,
                                   int32_t* out_col, float* out_val, dh_stream_t stream) {
  if (n < 0 || k < 0) return dh::fail(DH_ERR_INVALID, "dh_knn_graph_to_csr: negative size");
  if (n == 0) return DH_OK;
  if (!knn_idx || !w || !rowptr || !out_col || !out_val) return dh::fail(DH_ERR_INVALID, "dh_knn_graph_to_csr: null pointer");
  hipLaunchKernelGGL(knn_to_csr_kernel, dim3(blocks_for(n)), dim3(256), 0, dh::as_stream(stream), n, k, knn_idx, w, rowptr, out_col, out_val);
  return dh::check_launch("dh_knn_graph_to_csr");
}

extern "C" int dh_csr_union_count(int64_t n_rows, const int32_t* rowptr_a, const int32_t* col_a, const int32_t* rowptr_b,
                                  const int32_t* col_b, int32_t* out_counts, dh_stream_t stream) {
  if (n_rows < 0) return dh::fail(DH_ERR_INVALID, "dh_csr_union_count: negative size");
  if (n_rows == 0) return DH_OK;
  if (!rowptr_a || !rowptr_b || !out_counts) return dh::fail(DH_ERR_INVALID, "dh_csr_union_count: null pointer");
  hipLaunchKernelGGL(union_count_kernel, dim3(blocks_for(n_rows)), dim3(256), 0, dh::as_stream(stream), n_rows, rowptr_a, col_a, rowptr_b, col_b, out_counts);
  return dh::check_launch("dh_csr_union_count");
}

extern "C" int dh_csr_fuzzy_union_fill(int64_t n_rows, const int32_t* rowptr_a, const int32_t* col_a, const float* val_a,
                                       const int32_t* rowptr_b, const int32_t* col_b, const float* val_b,
                                       const int32_t* out_rowptr, int32_t* out_col, float* out_val, dh_stream_t stream) {
  if (n_rows < 0) return dh::fail(DH_ERR_INVALID, "dh_csr_fuzzy_union_fill: negative size");
  if (n_rows == 0) return DH_OK;
  if (!rowptr_a || !rowptr_b || !out_rowptr) return dh::fail(DH_ERR_INVALID, "dh_csr_fuzzy_union_fill: null pointer");
  hipLaunchKernelGGL(fuzzy_union_fill_kernel, dim3(blocks_for(n_rows)), dim3(256), 0, dh::as_stream(stream), n_rows, rowptr_a, col_a, val_a, rowptr_b, col_b, val_b, out_rowptr, out_col, out_val);
  return dh::check_launch("dh_csr_fuzzy_union_fill");
}
