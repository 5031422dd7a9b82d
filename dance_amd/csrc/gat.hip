// Edge softmax for graph attention: STAGATE's GATConv
// (dance/modules/spatial/spatial_domain/stagate.py:31-128: message() = softmax_over_in_edges(sigmoid(a_src[j] + a_dst[i])))
// and the standard GAT form (leaky_relu instead of sigmoid; scgnn2.py:1091-1118).
//
// The attention logit of edge (j -> i) depends on two per-node scalars only, so the [E, H, C] message tensor of the
// MessagePassing formulation never has to exist: this kernel turns (a_src, a_dst) into the normalised per-edge coefficient
// att[e] (one wavefront per destination row: max, sum, normalise — three passes over the row's <= few hundred edges, all
// in registers / L1), and the weighted aggregation out[i] = sum_e att[e] x[src(e)] is the ordinary CSR SpMM
// (dh_spmm_csr_f32 with val = att).  The backward kernel applies the softmax and activation Jacobians per row:
//   dt[e] = act'(t_e) * att[e] * (datt[e] - sum_k att[k] datt[k]),   t_e = a_src[src(e)] + a_dst[i]
// with datt[e] = <dOut[i], x[src(e)]> from dh_sddmm_csr_f32; it also emits the row sums of dt (= d a_dst).
#include "common.h"

namespace {

__device__ __forceinline__ float act_fwd(float t, int act, float slope) {
  if (act == 0) return 1.f / (1.f + expf(-t));        // sigmoid (STAGATE)
  return t > 0.f ? t : slope * t;                     // leaky_relu (GAT)
}
__device__ __forceinline__ float act_grad(float t, int act, float slope) {
  if (act == 0) {
    const float s = 1.f / (1.f + expf(-t));
    return s * (1.f - s);
  }
  return t > 0.f ? 1.f : slope;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void edge_softmax_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const float* __restrict__ a_src, const float* __restrict__ a_dst, int act, float slope,
                                                           const float* __restrict__ shift, float* __restrict__ att) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int s = rowptr[row], t = rowptr[row + 1];
  const float ad = a_dst ? a_dst[row] : 0.f;
  float m;
  if (shift) {
    // scGNN2's GATLayer (scgnn2.py:1071-1085) subtracts the GLOBAL maximum of all edge scores, not the row's: together with
    // the + 1e-16 of the denominator that is a different function for rows whose scores sit far below the global maximum
    // (their attentions shrink towards 0), so the caller hands that maximum in
    m = *shift;
  } else {
    m = -INFINITY;
    for (int e = s + lane; e < t; e += 64) m = fmaxf(m, act_fwd(a_src[col[e]] + ad, act, slope));
    m = wave_max(m);
  }
  float sum = 0.f;
  for (int e = s + lane; e < t; e += 64) {
    const float p = expf(act_fwd(a_src[col[e]] + ad, act, slope) - m);
    att[e] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / (sum + 1e-16f);  // torch_geometric.utils.softmax adds 1e-16 to the denominator
  for (int e = s + lane; e < t; e += 64) att[e] *= inv;
}

__global__ __launch_bounds__(256) void edge_softmax_bwd_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               const float* __restrict__ a_src, const float* __restrict__ a_dst, int act,
                                                               float slope, const float* __restrict__ att, const float* __restrict__ datt,
                                                               float* __restrict__ dt, float* __restrict__ d_a_dst) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int s = rowptr[row], t = rowptr[row + 1];
  const float ad = a_dst ? a_dst[row] : 0.f;
  float dot = 0.f;
  for (int e = s + lane; e < t; e += 64) dot += att[e] * datt[e];
  dot = wave_sum(dot);
  float rs = 0.f;
  for (int e = s + lane; e < t; e += 64) {
    const float g = act_grad(a_src[col[e]] + ad, act, slope) * att[e] * (datt[e] - dot);
    dt[e] = g;
    rs += g;
  }
  rs = wave_sum(rs);
  if (lane == 0 && d_a_dst) d_a_dst[row] = rs;
}

}  // namespace

extern "C" int dh_edge_softmax_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst, int act,
                                   float negative_slope, float* att, dh_stream_t stream) {
  return dh_edge_softmax_shift_f32(n_rows, rowptr, col, a_src, a_dst, act, negative_slope, nullptr, att, stream);
}

extern "C" int dh_edge_softmax_shift_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                                         int act, float negative_slope, const float* shift, float* att, dh_stream_t stream) {
  if (n_rows < 0) return dh::fail(DH_ERR_INVALID, "dh_edge_softmax_f32: negative size");
  if (n_rows == 0) return DH_OK;
  if (!rowptr || !col || !a_src || !att) return dh::fail(DH_ERR_INVALID, "dh_edge_softmax_f32: null pointer");
  if (act != 0 && act != 1) return dh::fail(DH_ERR_INVALID, "dh_edge_softmax_f32: act must be 0 (sigmoid) or 1 (leaky_relu)");
  hipLaunchKernelGGL(edge_softmax_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream), n_rows, rowptr, col, a_src, a_dst,
                     act, negative_slope, shift, att);
  return dh::check_launch("dh_edge_softmax_f32");
}

extern "C" int dh_edge_softmax_backward_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                                            int act, float negative_slope, const float* att, const float* datt, float* dt, float* d_a_dst,
                                            dh_stream_t stream) {
  if (n_rows < 0) return dh::fail(DH_ERR_INVALID, "dh_edge_softmax_backward_f32: negative size");
  if (n_rows == 0) return DH_OK;
  if (!rowptr || !col || !a_src || !att || !datt || !dt) return dh::fail(DH_ERR_INVALID, "dh_edge_softmax_backward_f32: null pointer");
  if (act != 0 && act != 1) return dh::fail(DH_ERR_INVALID, "dh_edge_softmax_backward_f32: bad act");
  hipLaunchKernelGGL(edge_softmax_bwd_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream), n_rows, rowptr, col, a_src,
                     a_dst, act, negative_slope, att, datt, dt, d_a_dst);
  return dh::check_launch("dh_edge_softmax_backward_f32");
}
