// The DEC heads' self-training target and KL loss (SimpleGCDEC / GC_DEC: dance/modules/spatial/spatial_domain/spagcn.py:398-425, 609-620;
// scDSC's clustering losses use the same target: dance/modules/single_modality/clustering/scdsc.py:431-441) as three kernels instead of the
// ~25 elementwise / reduction launches torch makes of them — at 500k spots x 10 clusters each of those is a ~10 us launch over 20 MB, and
// together they were a third of a SpaGCN iteration (BENCH_r05 c5 row: 0.19 - 0.32 ms of 0.73 - 0.85).
//
//   dh_dec_target_f32      : p_ij = (q_ij^2 / f_j) / sum_j' (q_ij'^2 / f_j'),  f = column sums of q (given: the caller all-reduces them
//                            when the spots are sharded)                                                      spagcn.py:421-425
//   dh_dec_kl_forward_f32  : loss = scale * sum_ij p_ij log(p_ij / (q_ij + eps)) — torch.mean over spots of the row sums when
//                            scale = 1 / n (:399-407); block partials in double, added in block order (deterministic)
//   dh_dec_kl_backward_f32 : dq_ij = -(g scale) p_ij / (q_ij + eps), g read from the device (the upstream gradient of the scalar loss)
//
// Arithmetic follows torch's: p / (q + eps) and its log in f32, the product in f32, the sum in double (torch's mean reduces in f32 with a
// tree; the result agrees to a few ulps, tests compare at 1e-6).
#include <algorithm>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void dec_target_kernel(int64_t n, int c, const float* __restrict__ q, int64_t ldq, const float* __restrict__ f,
                                                         float* __restrict__ p, int64_t ldp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* __restrict__ qi = q + i * ldq;
  float* __restrict__ pi = p + i * ldp;
  float s = 0.f;
  for (int j = 0; j < c; ++j) {  // q**2 / sum(q, 0), then the row sum in column order (torch.sum over a row of <= 64 entries: one thread, same order)
    const float v = qi[j];
    const float w = (v * v) / f[j];
    pi[j] = w;
    s += w;
  }
  for (int j = 0; j < c; ++j) pi[j] = pi[j] / s;
}

__global__ __launch_bounds__(256) void dec_kl_partial_kernel(int64_t total, int c, const float* __restrict__ p, int64_t ldp, const float* __restrict__ q,
                                                             int64_t ldq, float eps, double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / c;
    const int j = (int)(e - i * c);
    const float pv = p[i * ldp + j], qv = q[i * ldq + j];
    acc += (double)(pv * logf(pv / (qv + eps)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void dec_kl_final_kernel(int n_blocks, const double* __restrict__ partial, double scale, float* __restrict__ loss) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < n_blocks; b += 256) acc += partial[b];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)(red[0] * scale);
}

__global__ __launch_bounds__(256) void dec_kl_backward_kernel(int64_t total, int c, const float* __restrict__ p, int64_t ldp, const float* __restrict__ q,
                                                              int64_t ldq, float eps, float scale, const float* __restrict__ g, float* __restrict__ dq,
                                                              int64_t ldd) {
  const float gs = -(g ? *g : 1.f) * scale;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / c;
    const int j = (int)(e - i * c);
    dq[i * ldd + j] = gs * (p[i * ldp + j] / (q[i * ldq + j] + eps));
  }
}

constexpr int kKlBlocks = 1024;

}  // namespace

extern "C" int dh_dec_target_f32(int64_t n, int64_t c, const float* q, int64_t ldq, const float* colsum_q, float* p, int64_t ldp, dh_stream_t stream) {
  const char* me = "dh_dec_target_f32";
  if (n < 0 || c < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n == 0 || c == 0) return DH_OK;
  if (c > 4096) return dh::fail(DH_ERR_INVALID, "%s: more than 4096 clusters", me);
  if (!q || !colsum_q || !p || ldq < c || ldp < c) return dh::fail(DH_ERR_INVALID, "%s: bad pointer / leading dimension", me);
  hipLaunchKernelGGL(dec_target_kernel, dim3((unsigned)dh::ceil_div(n, 256)), dim3(256), 0, dh::as_stream(stream), n, (int)c, q, ldq, colsum_q, p, ldp);
  return dh::check_launch(me);
}

extern "C" size_t dh_dec_kl_workspace_bytes(void) { return (size_t)kKlBlocks * sizeof(double); }

extern "C" int dh_dec_kl_forward_f32(int64_t n, int64_t c, const float* p, int64_t ldp, const float* q, int64_t ldq, float eps, double scale, float* loss,
                                     void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_dec_kl_forward_f32";
  if (n < 0 || c < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (!loss) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  if (n > 0 && c > 0 && (!p || !q || ldp < c || ldq < c)) return dh::fail(DH_ERR_INVALID, "%s: bad pointer / leading dimension", me);
  if (!workspace || workspace_bytes < dh_dec_kl_workspace_bytes()) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace too small", me);
  hipStream_t st = dh::as_stream(stream);
  const int64_t total = n * c;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kKlBlocks, dh::ceil_div(total, 1024)));
  double* partial = static_cast<double*>(workspace);
  hipLaunchKernelGGL(dec_kl_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, total, (int)std::max<int64_t>(c, 1), p, ldp, q, ldq, eps, partial);
  hipLaunchKernelGGL(dec_kl_final_kernel, dim3(1), dim3(256), 0, st, nb, partial, scale, loss);
  return dh::check_launch(me);
}

extern "C" int dh_dec_kl_backward_f32(int64_t n, int64_t c, const float* p, int64_t ldp, const float* q, int64_t ldq, float eps, double scale, const float* g,
                                      float* dq, int64_t ldd, dh_stream_t stream) {
  const char* me = "dh_dec_kl_backward_f32";
  if (n < 0 || c < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n == 0 || c == 0) return DH_OK;
  if (!p || !q || !dq || ldp < c || ldq < c || ldd < c) return dh::fail(DH_ERR_INVALID, "%s: bad pointer / leading dimension", me);
  const int64_t total = n * c;
  const unsigned grid = (unsigned)std::min<int64_t>(dh::ceil_div(total, 256), 8192);
  hipLaunchKernelGGL(dec_kl_backward_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), total, (int)c, p, ldp, q, ldq, eps, (float)scale, g, dq, ldd);
  return dh::check_launch(me);
}
