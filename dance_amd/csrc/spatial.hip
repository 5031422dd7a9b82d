// kNN-truncated Gaussian spatial adjacency for SpaGCN at scale (SURVEY.md §0.5, §8b export list):
// the reference multiplies by the DENSE kernel exp(-d_ij^2 / (2 l^2)) of all spot pairs
// (dance/modules/spatial/spatial_domain/spagcn.py:249-251,807-809); beyond a few tens of thousands of spots that matrix
// does not exist, and the kernel's tail is negligible: dh_spatial_gaussian_knn keeps each spot's k nearest spots
// (exact search, self included: dh_knn_bruteforce_f32) and evaluates the same fp32 kernel expression on them, writing a
// CSR whose rows are sorted by column — ready for dh_spmm_csr_f32.  l <= 0 writes the distances themselves (the form
// SpaGCN.search_l / calc_adj_exp consume).
#include "common.h"

extern "C" size_t dh_knn_bruteforce_f32_workspace_bytes(int64_t n, int64_t d, int64_t n_queries, int k, int algo);
extern "C" int dh_knn_bruteforce_f32(int64_t n, int64_t d, const float* X, int64_t ldx, int64_t q_begin, int64_t q_end, int k, int algo,
                                     int32_t* out_idx, float* out_dist, void* workspace, size_t workspace_bytes, dh_stream_t stream);

namespace {

// one thread per spot: order its k (index, distance) pairs by index (k <= 64: insertion sort in registers / scratch)
__global__ __launch_bounds__(256) void knn_rows_to_csr_kernel(int64_t n, int k, const int32_t* __restrict__ idx, const float* __restrict__ dist,
                                                              float denom, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                              float* __restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    rowptr[n] = (int32_t)(n * k);
    return;
  }
  rowptr[i] = (int32_t)(i * k);
  const int32_t* ii = idx + i * k;
  const float* di = dist + i * k;
  int32_t* oc = col + i * k;
  float* ov = val + i * k;
  for (int j = 0; j < k; ++j) {  // rank of entry j among the row's indices (distinct: a spot is listed once)
    const int32_t c = ii[j];
    int rank = 0;
    for (int t = 0; t < k; ++t) rank += (ii[t] < c) || (ii[t] == c && t < j);
    const float v = di[j];
    oc[rank] = c;
    ov[rank] = denom > 0.f ? expf(__fdiv_rn(-(v * v), denom)) : v;
  }
}

}  // namespace

extern "C" size_t dh_spatial_gaussian_knn_workspace_bytes(int64_t n, int64_t d, int k) {
  if (n <= 0 || d <= 0 || k <= 0) return 0;
  const size_t lists = (((size_t)n * k * 4 + 255) & ~(size_t)255) * 2;
  return lists + dh_knn_bruteforce_f32_workspace_bytes(n, d, n, k, 0) + 256;
}

extern "C" int dh_spatial_gaussian_knn(int64_t n, int64_t d, const float* X, int64_t ldx, int k, double l, int32_t* out_rowptr,
                                       int32_t* out_col, float* out_val, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_spatial_gaussian_knn";
  if (n < 0 || d <= 0 || k <= 0) return dh::fail(DH_ERR_INVALID, "%s: bad size", me);
  if (!out_rowptr) return dh::fail(DH_ERR_INVALID, "%s: null out_rowptr", me);
  hipStream_t st = dh::as_stream(stream);
  if (n == 0) return dh::zero_async(out_rowptr, sizeof(int32_t), st) == hipSuccess ? DH_OK : dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me);
  if (!X || !out_col || !out_val) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (k > n || k > 64) return dh::fail(DH_ERR_INVALID, "%s: k must be <= min(n, 64)", me);
  if (n * (int64_t)k >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: n * k >= 2^31", me);
  if (!workspace || workspace_bytes < dh_spatial_gaussian_knn_workspace_bytes(n, d, k)) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace too small", me);
  char* ws = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t list_bytes = ((size_t)n * k * 4 + 255) & ~(size_t)255;
  int32_t* idx = reinterpret_cast<int32_t*>(ws);
  float* dist = reinterpret_cast<float*>(ws + list_bytes);
  char* knn_ws = ws + 2 * list_bytes;
  const size_t knn_bytes = dh_knn_bruteforce_f32_workspace_bytes(n, d, n, k, 0);
  int rc = dh_knn_bruteforce_f32(n, d, X, ldx, 0, n, k, 0, idx, dist, knn_ws, knn_bytes, stream);
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(knn_rows_to_csr_kernel, dim3((unsigned)dh::ceil_div(n + 1, 256)), dim3(256), 0, st, n, k, idx, dist,
                     l > 0 ? (float)(2.0 * (l * l)) : 0.f, out_rowptr, out_col, out_val);
  return dh::check_launch(me);
}
