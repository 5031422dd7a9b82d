// Small streaming kernels of the layers' backward pass (HBM-bound, one touch per element).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// G = dY where Y > 0 else 0  (autograd of F.relu; threshold_backward semantics: Y <= 0 -> 0)
template <bool VEC4>
__global__ __launch_bounds__(256) void relu_backward_kernel(int64_t n_rows, int64_t width,
                                                            const float* __restrict__ Y, int64_t ldy,
                                                            const float* __restrict__ dY, int64_t lddy,
                                                            float* __restrict__ G, int64_t ldg) {
  if constexpr (VEC4) {
    const int64_t wv = width / 4, total = n_rows * wv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / wv, c = (i % wv) * 4;
      const f32x4 y = *reinterpret_cast<const f32x4*>(Y + r * ldy + c);
      const f32x4 d = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dY + r * lddy + c));
      f32x4 g;
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.f ? d[k] : 0.f;
      *reinterpret_cast<f32x4*>(G + r * ldg + c) = g;
    }
  } else {
    const int64_t total = n_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / width, c = i % width;
      G[r * ldg + c] = Y[r * ldy + c] > 0.f ? dY[r * lddy + c] : 0.f;
    }
  }
}

// Column sums, pass 1: block b sums rows [b*rows_per_block, ...) -> partial[b][width].  A group of G lanes (G = 64
// for wide matrices, 64 / 32 / 16 for narrow ones so that a wavefront covers several rows) walks the columns of a
// row with coalesced loads; the block's row groups run in parallel and are combined through LDS in a fixed order.
template <int G>
__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t n_rows, int64_t width,
                                                             const float* __restrict__ X, int64_t ldx,
                                                             int64_t rows_per_block,
                                                             float* __restrict__ partial) {
  constexpr int NG = 256 / G;  // row groups per block
  __shared__ float red[256];
  const int g = threadIdx.x % G, rg = threadIdx.x / G;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(n_rows, r0 + rows_per_block);
  for (int64_t c = (int64_t)blockIdx.y * G + g; c < (int64_t)(blockIdx.y + 1) * G; c += G) {
    float s = 0.f;
    if (c < width) {
      int64_t r = r0 + rg;
      for (; r + 3 * NG < r1; r += 4 * NG) {  // 4 independent loads in flight
        const float a = X[r * ldx + c], b = X[(r + NG) * ldx + c], d = X[(r + 2 * NG) * ldx + c], e = X[(r + 3 * NG) * ldx + c];
        s += a; s += b; s += d; s += e;
      }
      for (; r < r1; r += NG) s += X[r * ldx + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && c < width) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NG; ++k) t += red[k * G + g];
      partial[(int64_t)blockIdx.x * width + c] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(int64_t n_blocks, int64_t width,
                                                           const float* __restrict__ partial,
                                                           float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  float s = 0.f;
  int64_t b = 0;
  for (; b + 8 <= n_blocks; b += 8) {  // 8 loads in flight, added in block order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(b + u) * width + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < n_blocks; ++b) s += partial[b * width + c];
  out[c] = s;
}

constexpr int64_t kColsumRows = 2048;

// out = a X + b Y, each product and the sum rounded separately (what the torch expression `a * X + b * Y` computes with three
// kernels and two temporaries); Y == nullptr: out = a X.
template <bool VEC>
__global__ __launch_bounds__(256) void axpby_kernel(int64_t n_rows, int64_t width, float a, const float* __restrict__ X, int64_t ldx, float b,
                                                    const float* __restrict__ Y, int64_t ldy, float* __restrict__ O, int64_t ldo) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int64_t per_row = VEC ? width / 4 : width, total = n_rows * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per_row, c = (i % per_row) * (VEC ? 4 : 1);
    if constexpr (VEC) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
      f32x4 o;
      if (Y) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(Y + r * ldy + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(__fmul_rn(a, x[j]), __fmul_rn(b, y[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a * x[j];
      }
      *reinterpret_cast<f32x4*>(O + r * ldo + c) = o;
    } else {
      const float x = X[r * ldx + c];
      O[r * ldo + c] = Y ? __fadd_rn(__fmul_rn(a, x), __fmul_rn(b, Y[r * ldy + c])) : a * x;
    }
  }
}

}  // namespace

extern "C" int dh_axpby_f32(int64_t n_rows, int64_t width, float a, const float* X, int64_t ldx, float b, const float* Y, int64_t ldy, float* out,
                            int64_t ldo, dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_axpby_f32: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!X || !out) return dh::fail(DH_ERR_INVALID, "dh_axpby_f32: null pointer");
  if (ldx < width || ldo < width || (Y && ldy < width)) return dh::fail(DH_ERR_INVALID, "dh_axpby_f32: leading dimension < width");
  const bool vec = width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!Y || ldy % 4 == 0) && dh::aligned16(X) && dh::aligned16(out) &&
                   (!Y || dh::aligned16(Y));
  const int64_t work = vec ? n_rows * (width / 4) : n_rows * width;
  const unsigned grid = (unsigned)(dh::ceil_div(work, 256) < 16384 ? dh::ceil_div(work, 256) : 16384);
  hipStream_t st = dh::as_stream(stream);
  if (vec)
    hipLaunchKernelGGL(axpby_kernel<true>, dim3(grid), dim3(256), 0, st, n_rows, width, a, X, ldx, b, Y, ldy, out, ldo);
  else
    hipLaunchKernelGGL(axpby_kernel<false>, dim3(grid), dim3(256), 0, st, n_rows, width, a, X, ldx, b, Y, ldy, out, ldo);
  return dh::check_launch("dh_axpby_f32");
}

extern "C" int dh_relu_backward_f32(int64_t n_rows, int64_t width, const float* Y, int64_t ldy,
                                    const float* dY, int64_t lddy, float* G, int64_t ldg,
                                    dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_relu_backward_f32: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!Y || !dY || !G) return dh::fail(DH_ERR_INVALID, "dh_relu_backward_f32: null pointer");
  if (ldy < width || lddy < width || ldg < width)
    return dh::fail(DH_ERR_INVALID, "dh_relu_backward_f32: leading dimension < width");
  hipStream_t st = dh::as_stream(stream);
  const bool vec = width % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && ldg % 4 == 0 &&
                   dh::aligned16(Y) && dh::aligned16(dY) && dh::aligned16(G);
  const int64_t work = vec ? n_rows * (width / 4) : n_rows * width;
  const unsigned grid = (unsigned)(dh::ceil_div(work, 256) < 8192 ? dh::ceil_div(work, 256) : 8192);
  if (vec)
    hipLaunchKernelGGL(relu_backward_kernel<true>, dim3(grid), dim3(256), 0, st, n_rows, width, Y, ldy, dY, lddy, G, ldg);
  else
    hipLaunchKernelGGL(relu_backward_kernel<false>, dim3(grid), dim3(256), 0, st, n_rows, width, Y, ldy, dY, lddy, G, ldg);
  return dh::check_launch("dh_relu_backward_f32");
}

// Rows per block of pass 1.  2048 for the tall matrices (1M rows: thousands of blocks anyway) and for mini-batch matrices (one block:
// its partial sums ARE the result, no second launch); in between — 8192 x 300, the bias gradients of graph-sc's large batches — 2048 rows
// per block meant 20 workgroups each walking 512 rows four at a time, 45 us for 10 MB (round 6: 11 ms of a 229 ms epoch): there the rows
// are dealt to ~2048 / column-blocks workgroups, at least 64 each.
static int64_t colsum_rows_per_block(int64_t n_rows, int64_t width) {
  if (n_rows * width <= 262144) return kColsumRows;
  const int64_t cb = width > 32 ? dh::ceil_div(width, 64) : width > 16 ? 1 : dh::ceil_div(width, 16);
  const int64_t nb_target = 2048 / cb > 1 ? 2048 / cb : 1;
  int64_t rpb = (dh::ceil_div(n_rows, nb_target) + 15) / 16 * 16;
  return rpb < 64 ? 64 : rpb > kColsumRows ? kColsumRows : rpb;
}

extern "C" size_t dh_colsum_f32_workspace_bytes(int64_t n_rows, int64_t width) {
  if (n_rows <= 0 || width <= 0) return 0;
  return (size_t)dh::ceil_div(n_rows, colsum_rows_per_block(n_rows, width)) * (size_t)width * sizeof(float);  // (>= the 2048-row layout dh_colsum_bf16 uses)
}

extern "C" int dh_colsum_f32(int64_t n_rows, int64_t width, const float* X, int64_t ldx, float* out,
                             void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_colsum_f32: negative size");
  if (width == 0) return DH_OK;
  if (!out) return dh::fail(DH_ERR_INVALID, "dh_colsum_f32: null out");
  hipStream_t st = dh::as_stream(stream);
  if (n_rows == 0) {
    if (dh::zero_async(out, width * sizeof(float), st) != hipSuccess)
      return dh::fail(DH_ERR_LAUNCH, "dh_colsum_f32: memset failed");
    return DH_OK;
  }
  if (!X || ldx < width) return dh::fail(DH_ERR_INVALID, "dh_colsum_f32: bad X/ldx");
  const size_t need = dh_colsum_f32_workspace_bytes(n_rows, width);
  if (!workspace || workspace_bytes < need)
    return dh::fail(DH_ERR_WORKSPACE, "dh_colsum_f32: workspace %zu < %zu bytes", workspace_bytes, need);
  const int64_t rpb = colsum_rows_per_block(n_rows, width);
  const int64_t nb = dh::ceil_div(n_rows, rpb);
  // one row block (a mini-batch: <= 2048 rows): its "partial" sums are the result — written straight to out, no second launch
  float* partial = nb == 1 ? out : static_cast<float*>(workspace);
  if (width > 32) {
    dim3 grid((unsigned)nb, (unsigned)dh::ceil_div(width, 64));
    hipLaunchKernelGGL(colsum_partial_kernel<64>, grid, dim3(256), 0, st, n_rows, width, X, ldx, rpb, partial);
  } else if (width > 16) {
    dim3 grid((unsigned)nb, 1);
    hipLaunchKernelGGL(colsum_partial_kernel<32>, grid, dim3(256), 0, st, n_rows, width, X, ldx, rpb, partial);
  } else {
    dim3 grid((unsigned)nb, (unsigned)dh::ceil_div(width, 16));
    hipLaunchKernelGGL(colsum_partial_kernel<16>, grid, dim3(256), 0, st, n_rows, width, X, ldx, rpb, partial);
  }
  if (nb > 1) hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)dh::ceil_div(width, 256)), dim3(256), 0, st, nb, width, partial, out);
  return dh::check_launch("dh_colsum_f32");
}

// X[i,:] = act(X[i,:] + bias) in place — epilogue of the dense layers (nn.Linear bias, GraphConvolution bias on a
// dense adjacency).
namespace {
__global__ __launch_bounds__(256) void bias_act_kernel(int64_t n_rows, int64_t width, float* __restrict__ X, int64_t ldx,
                                                       const float* __restrict__ bias, int act) {
  const int64_t total = n_rows * width;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / width, c = i % width;
    float v = X[r * ldx + c] + (bias ? bias[c] : 0.f);
    X[r * ldx + c] = act == DH_ACT_RELU ? fmaxf(v, 0.f) : v;
  }
}
}  // namespace

extern "C" int dh_bias_act_f32(int64_t n_rows, int64_t width, float* X, int64_t ldx, const float* bias, int act,
                               dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_bias_act_f32: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!X || ldx < width) return dh::fail(DH_ERR_INVALID, "dh_bias_act_f32: bad X/ldx");
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_bias_act_f32: bad act %d", act);
  const int64_t work = n_rows * width;
  const unsigned grid = (unsigned)(dh::ceil_div(work, 256) < 8192 ? dh::ceil_div(work, 256) : 8192);
  hipLaunchKernelGGL(bias_act_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), n_rows, width, X, ldx, bias, act);
  return dh::check_launch("dh_bias_act_f32");
}

// SpaGCN's Gaussian kernel (spagcn.py:249-251,807-809): e = exp(-(d*d) / (2 l^2)) in f32 like numpy evaluates it
// (f32 square, f32 negate, divide by the f32-rounded scalar 2 l^2, expf).  Optionally writes e and/or per-row sums
// (f64 accumulation, one wavefront per row): calculate_p needs only the sums, so search_l's bisection streams the
// N^2 distances once per step and writes N floats.
namespace {
__global__ __launch_bounds__(256) void gaussian_kernel_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ D,
                                                              int64_t ldd, float denom, float* __restrict__ out, int64_t ldo,
                                                              float* __restrict__ rowsum) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* d = D + row * ldd;
  double acc = 0.0;
  for (int64_t c = lane; c < n_cols; c += 64) {
    const float v = d[c];
    const float e = expf(__fdiv_rn(-(v * v), denom));
    if (out) out[row * ldo + c] = e;
    acc += (double)e;
  }
  if (rowsum) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) rowsum[row] = (float)acc;
  }
}
}  // namespace

extern "C" int dh_gaussian_kernel_f32(int64_t n_rows, int64_t n_cols, const float* D, int64_t ldd, double l, float* out,
                                      int64_t ldo, float* rowsum, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_gaussian_kernel_f32: negative size");
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!D || ldd < n_cols || (out && ldo < n_cols)) return dh::fail(DH_ERR_INVALID, "dh_gaussian_kernel_f32: bad pointer / leading dimension");
  if (!(l > 0)) return dh::fail(DH_ERR_INVALID, "dh_gaussian_kernel_f32: l must be positive");
  hipLaunchKernelGGL(gaussian_kernel_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream),
                     n_rows, n_cols, D, ldd, (float)(2.0 * (l * l)), out, ldo, rowsum);
  return dh::check_launch("dh_gaussian_kernel_f32");
}

// Dense part of the inner-product decoder loss of graph-sc (dance/modules/single_modality/clustering/graphsc.py:208-216:
// binary_cross_entropy_with_logits(z z^T, adj, pos_weight) over a B x B block whose target `adj` is all zero except for the
// few edges among the batch's own cells).  With y = 0 the element loss is softplus(x) and its derivative sigmoid(x); the
// sparse y = 1 corrections are applied by the caller on the edge list.  torch evaluates this as ~15 elementwise passes over
// the 268 MB logit matrix (B = 8192) plus a dense target; here the forward is ONE read (row sums of softplus, f64
// accumulation, one wavefront per row — deterministic) and the backward one read + one write.
namespace {
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void softplus_rowsum_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                              float* __restrict__ rowsum) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* x = X + row * ldx;
  double acc = 0.0;
  for (int64_t c = lane; c < n_cols; c += 64) acc += (double)softplus_f(x[c]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) rowsum[row] = (float)acc;
}

__global__ __launch_bounds__(256) void sigmoid_scale_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ scale, float* __restrict__ out, int64_t ldo) {
  const float s = scale[0];
  const int64_t total = n_rows * n_cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n_cols, c = i - r * n_cols;
    const float x = X[r * ldx + c];
    out[r * ldo + c] = s / (1.f + expf(-x));
  }
}
}  // namespace

extern "C" int dh_softplus_rowsum_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, float* rowsum, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_softplus_rowsum_f32: negative size");
  if (n_rows == 0) return DH_OK;
  if (!rowsum || (n_cols > 0 && (!X || ldx < n_cols))) return dh::fail(DH_ERR_INVALID, "dh_softplus_rowsum_f32: bad pointer / leading dimension");
  hipLaunchKernelGGL(softplus_rowsum_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx, rowsum);
  return dh::check_launch("dh_softplus_rowsum_f32");
}

extern "C" int dh_sigmoid_scale_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const float* scale, float* out, int64_t ldo,
                                    dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_sigmoid_scale_f32: negative size");
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!X || !out || !scale || ldx < n_cols || ldo < n_cols) return dh::fail(DH_ERR_INVALID, "dh_sigmoid_scale_f32: bad pointer / leading dimension");
  const int64_t work = n_rows * n_cols;
  const unsigned grid = (unsigned)(dh::ceil_div(work, 256) < 65536 ? dh::ceil_div(work, 256) : 65536);
  hipLaunchKernelGGL(sigmoid_scale_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx, scale, out, ldo);
  return dh::check_launch("dh_sigmoid_scale_f32");
}

// Count-matrix normalisation feeding the graph builders: the arithmetic of scanpy's normalize_total /
// log1p / scale, which the reference pipelines call through AnnDataTransform (scdsc.py:113-131, sctag.py:119-139,
// dance/transforms/normalize.py:531-679), as streaming kernels over the dense cell x gene matrix.
//   dh_rowsum_masked_f32     : out[r] = sum_c X[r,c] over the columns with colmask[c] != 0 (NULL: all) — counts per cell,
//                              optionally without the "highly expressed" genes; f64 accumulation, one wavefront per row
//   dh_rowscale_log1p_f32    : out[r,c] = g(X[r,c] / divisor[r]), g = identity or log1p(.) / ln(base)  (a true division, as numpy's)
//   dh_col_standardize_f32   : out[r,c] = clip((X[r,c] - mean[c]) / std[c], +-max_value)  (f64 statistics; max_value <= 0: no clipping)
//   dh_col_moments_f32       : per-row-block f64 column sums of X and of fl32(X^2) (mean / variance of every gene)
//   dh_col_any_gt_f32        : flag[c] = any_r X[r,c] > thresh[r]   (the "highly expressed in some cell" test)
namespace {
__global__ __launch_bounds__(256) void rowsum_masked_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                            const uint8_t* __restrict__ colmask, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* x = X + row * ldx;
  double acc = 0.0;
  for (int64_t c = lane; c < n_cols; c += 64)
    if (!colmask || colmask[c]) acc += (double)x[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) out[row] = (float)acc;
}

__global__ __launch_bounds__(256) void rowscale_log1p_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ factor, int do_log1p, float inv_log_base,
                                                             float* __restrict__ out, int64_t ldo) {
  const int64_t total = n_rows * n_cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n_cols, c = i - r * n_cols;
    float v = X[r * ldx + c];
    if (factor) v = v / factor[r];
    if (do_log1p) v = log1pf(v) * inv_log_base;
    out[r * ldo + c] = v;
  }
}

__global__ __launch_bounds__(256) void col_standardize_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                              const double* __restrict__ mean, const double* __restrict__ std, float max_value,
                                                              float* __restrict__ out, int64_t ldo) {
  const int64_t total = n_rows * n_cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n_cols, c = i - r * n_cols;
    float v = X[r * ldx + c];
    if (mean) v = (float)((double)v - mean[c]);  // numpy's in-place "X -= mean" with a float64 mean: f64 arithmetic, f32 store
    v = (float)((double)v / std[c]);
    if (max_value > 0.f) {
      v = fminf(v, max_value);
      if (mean) v = fmaxf(v, -max_value);
    }
    out[r * ldo + c] = v;
  }
}

// partial[b][0][c] = sum over the block's rows of X[r,c], partial[b][1][c] = sum of fl32(X[r,c]^2), both in f64
// (numpy's mean(X, dtype=f64) and mean(multiply(X, X), dtype=f64)); the caller adds the blocks in order.
__global__ __launch_bounds__(256) void col_moments_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                          int64_t rows_per_block, double* __restrict__ partial) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
  double s = 0.0, q = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    const float x = X[r * ldx + c];
    s += (double)x;
    q += (double)(x * x);
  }
  partial[((int64_t)blockIdx.y * 2 + 0) * n_cols + c] = s;
  partial[((int64_t)blockIdx.y * 2 + 1) * n_cols + c] = q;
}

// flag[c] = 1 if X[r,c] > thresh[r] for any row r (flag zero-initialised by the launcher; all writers store the same value)
__global__ __launch_bounds__(256) void col_any_gt_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ thresh, uint8_t* __restrict__ flag) {
  const int64_t total = n_rows * n_cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n_cols, c = i - r * n_cols;
    if (X[r * ldx + c] > thresh[r]) flag[c] = 1;
  }
}
unsigned flat_grid(int64_t work) { return (unsigned)(dh::ceil_div(work, 256) < 65536 ? dh::ceil_div(work, 256) : 65536); }
}  // namespace

extern "C" int dh_rowsum_masked_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const uint8_t* colmask, float* out,
                                    dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_rowsum_masked_f32: negative size");
  if (n_rows == 0) return DH_OK;
  if (!out || (n_cols > 0 && (!X || ldx < n_cols))) return dh::fail(DH_ERR_INVALID, "dh_rowsum_masked_f32: bad pointer / leading dimension");
  hipLaunchKernelGGL(rowsum_masked_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx, colmask, out);
  return dh::check_launch("dh_rowsum_masked_f32");
}

extern "C" int dh_rowscale_log1p_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const float* factor, int do_log1p,
                                     double log_base, float* out, int64_t ldo, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_rowscale_log1p_f32: negative size");
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!X || !out || ldx < n_cols || ldo < n_cols) return dh::fail(DH_ERR_INVALID, "dh_rowscale_log1p_f32: bad pointer / leading dimension");
  const float inv = (do_log1p && log_base > 0) ? (float)(1.0 / log(log_base)) : 1.f;
  hipLaunchKernelGGL(rowscale_log1p_kernel, dim3(flat_grid(n_rows * n_cols)), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx, factor,
                     do_log1p, inv, out, ldo);
  return dh::check_launch("dh_rowscale_log1p_f32");
}

extern "C" int dh_col_standardize_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const double* mean, const double* std,
                                      double max_value, float* out, int64_t ldo, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_col_standardize_f32: negative size");
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!X || !out || !std || ldx < n_cols || ldo < n_cols) return dh::fail(DH_ERR_INVALID, "dh_col_standardize_f32: bad pointer / leading dimension");
  hipLaunchKernelGGL(col_standardize_kernel, dim3(flat_grid(n_rows * n_cols)), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx, mean, std,
                     (float)max_value, out, ldo);
  return dh::check_launch("dh_col_standardize_f32");
}

extern "C" int dh_col_moments_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, int64_t rows_per_block, double* partial,
                                  dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || rows_per_block <= 0) return dh::fail(DH_ERR_INVALID, "dh_col_moments_f32: bad size");
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!X || !partial || ldx < n_cols) return dh::fail(DH_ERR_INVALID, "dh_col_moments_f32: bad pointer / leading dimension");
  const int64_t nb = dh::ceil_div(n_rows, rows_per_block);
  if (nb > 65535) return dh::fail(DH_ERR_INVALID, "dh_col_moments_f32: more than 65535 row blocks (raise rows_per_block)");
  hipLaunchKernelGGL(col_moments_kernel, dim3((unsigned)dh::ceil_div(n_cols, 256), (unsigned)nb), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx,
                     rows_per_block, partial);
  return dh::check_launch("dh_col_moments_f32");
}

extern "C" int dh_col_any_gt_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const float* thresh, uint8_t* flag,
                                 dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_col_any_gt_f32: negative size");
  if (n_cols == 0) return DH_OK;
  if (!flag) return dh::fail(DH_ERR_INVALID, "dh_col_any_gt_f32: null flag");
  hipStream_t st = dh::as_stream(stream);
  if (dh::zero_async(flag, (size_t)n_cols, st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "dh_col_any_gt_f32: memset failed");
  if (n_rows == 0) return DH_OK;
  if (!X || !thresh || ldx < n_cols) return dh::fail(DH_ERR_INVALID, "dh_col_any_gt_f32: bad pointer / leading dimension");
  hipLaunchKernelGGL(col_any_gt_kernel, dim3(flat_grid(n_rows * n_cols)), dim3(256), 0, st, n_rows, n_cols, X, ldx, thresh, flag);
  return dh::check_launch("dh_col_any_gt_f32");
}
