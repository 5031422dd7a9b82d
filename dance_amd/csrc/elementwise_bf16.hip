// bf16-storage variants of the layer-backward helpers (config C3): ReLU backward and the bias-gradient column sum.
// All arithmetic is fp32; bf16 values are widened exactly.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float widen(uint16_t h) { return __uint_as_float((unsigned int)h << 16); }

// G[i] = dY[i] where Y[i] > 0 else 0; 8 bf16 (16 bytes) per lane when rows allow
template <bool VEC>
__global__ __launch_bounds__(256) void relu_backward_bf16_kernel(int64_t n_rows, int64_t width, const uint16_t* __restrict__ Y,
                                                                 int64_t ldy, const uint16_t* __restrict__ dY, int64_t lddy,
                                                                 uint16_t* __restrict__ G, int64_t ldg) {
  if constexpr (VEC) {
    const int64_t vw = width / 8, total = n_rows * vw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / vw, c = (i % vw) * 8;
      const u32x4 y = *reinterpret_cast<const u32x4*>(Y + r * ldy + c);
      u32x4 d = *reinterpret_cast<const u32x4*>(dY + r * lddy + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // bf16 > 0: sign bit clear and not zero (a NaN in Y passes dY through, like torch's (y > 0) mask does not — keep the mask semantics: NaN > 0 is false)
        const unsigned int lo = y[k] & 0xffffu, hi = y[k] >> 16;
        const bool plo = lo != 0 && lo < 0x7f81u, phi = hi != 0 && hi < 0x7f81u;
        d[k] = (plo ? (d[k] & 0xffffu) : 0u) | (phi ? (d[k] & 0xffff0000u) : 0u);
      }
      *reinterpret_cast<u32x4*>(G + r * ldg + c) = d;
    }
  } else {
    const int64_t total = n_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / width, c = i % width;
      G[r * ldg + c] = widen(Y[r * ldy + c]) > 0.f ? dY[r * lddy + c] : (uint16_t)0;
    }
  }
}

// per-block partial column sums; G lanes across columns, 256 / G row groups (as colsum_partial_kernel, elementwise.hip)
template <int G>
__global__ __launch_bounds__(256) void colsum_bf16_partial_kernel(int64_t n_rows, int64_t width, const uint16_t* __restrict__ X,
                                                                  int64_t ldx, int64_t rows_per_block, float* __restrict__ partial) {
  constexpr int NG = 256 / G;
  __shared__ float red[256];
  const int g = threadIdx.x % G, rg = threadIdx.x / G;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
  const int64_t c = (int64_t)blockIdx.y * G + g;
  float s = 0.f;
  if (c < width) {
    int64_t r = r0 + rg;
    for (; r + 3 * NG < r1; r += 4 * NG) {
      const float a = widen(X[r * ldx + c]), b = widen(X[(r + NG) * ldx + c]), d = widen(X[(r + 2 * NG) * ldx + c]),
                  e = widen(X[(r + 3 * NG) * ldx + c]);
      s += a; s += b; s += d; s += e;
    }
    for (; r < r1; r += NG) s += widen(X[r * ldx + c]);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rg == 0 && c < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) t += red[k * G + g];
    partial[(int64_t)blockIdx.x * width + c] = t;
  }
}

// The same partial sums for 16-byte aligned rows whose width is a multiple of 8: a lane takes 8 consecutive columns (one 16-byte load
// instead of a 2-byte one: the form above moves 128 bytes per wave instruction), 8 lanes = 64 columns = one 128-byte line of a row,
// 32 row groups per block, 8 rows in flight per lane.  At the shapes of a scDeepSort batch (65536 x 200) the launch is a chain of
// dependent round trips — 2048 rows per block by the workspace contract — and the chain is 8 batches long instead of 128.
__global__ __launch_bounds__(256) void colsum_bf16_partial_vec_kernel(int64_t n_rows, int64_t width, const uint16_t* __restrict__ X, int64_t ldx,
                                                                      int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ float red[32][65];
  const int cc = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
  const int64_t c = (int64_t)blockIdx.y * 64 + 8 * cc;
  const bool live = c < width;
  const uint16_t* base = X + (live ? c : 0);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  for (int64_t r = r0 + rg; r < r1; r += 32 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + min(r + 32 * u, r1 - 1) * ldx);  // clamped: no branch between the loads
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = r + 32 * u < r1;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s[2 * w] += in ? __uint_as_float(v[u][w] << 16) : 0.f;
        s[2 * w + 1] += in ? __uint_as_float(v[u][w] & 0xffff0000u) : 0.f;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rg][8 * cc + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64 && (int64_t)blockIdx.y * 64 + threadIdx.x < width) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
    partial[(int64_t)blockIdx.x * width + (int64_t)blockIdx.y * 64 + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void colsum_bf16_final_kernel(int64_t n_blocks, int64_t width, const float* __restrict__ partial,
                                                                float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  float s = 0.f;
  for (int64_t b = 0; b < n_blocks; ++b) s += partial[b * width + c];
  out[c] = s;
}

constexpr int64_t kColsumRows = 2048;

}  // namespace

extern "C" int dh_relu_backward_bf16(int64_t n_rows, int64_t width, const uint16_t* Y, int64_t ldy, const uint16_t* dY,
                                     int64_t lddy, uint16_t* G, int64_t ldg, dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_relu_backward_bf16: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!Y || !dY || !G) return dh::fail(DH_ERR_INVALID, "dh_relu_backward_bf16: null pointer");
  if (ldy < width || lddy < width || ldg < width) return dh::fail(DH_ERR_INVALID, "dh_relu_backward_bf16: leading dimension < width");
  hipStream_t st = dh::as_stream(stream);
  const bool vec = width % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && ldg % 8 == 0 && dh::aligned16(Y) && dh::aligned16(dY) && dh::aligned16(G);
  const int64_t work = vec ? n_rows * (width / 8) : n_rows * width;
  const unsigned grid = (unsigned)(dh::ceil_div(work, 256) < 8192 ? dh::ceil_div(work, 256) : 8192);
  if (vec) hipLaunchKernelGGL(relu_backward_bf16_kernel<true>, dim3(grid), dim3(256), 0, st, n_rows, width, Y, ldy, dY, lddy, G, ldg);
  else hipLaunchKernelGGL(relu_backward_bf16_kernel<false>, dim3(grid), dim3(256), 0, st, n_rows, width, Y, ldy, dY, lddy, G, ldg);
  return dh::check_launch("dh_relu_backward_bf16");
}

extern "C" int dh_colsum_bf16(int64_t n_rows, int64_t width, const uint16_t* X, int64_t ldx, float* out, void* workspace,
                              size_t workspace_bytes, dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_colsum_bf16: negative size");
  if (width == 0) return DH_OK;
  if (!out) return dh::fail(DH_ERR_INVALID, "dh_colsum_bf16: null out");
  hipStream_t st = dh::as_stream(stream);
  if (n_rows == 0) {
    if (dh::zero_async(out, width * sizeof(float), st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "dh_colsum_bf16: memset failed");
    return DH_OK;
  }
  if (!X || ldx < width) return dh::fail(DH_ERR_INVALID, "dh_colsum_bf16: bad X/ldx");
  const int64_t nb = dh::ceil_div(n_rows, kColsumRows);
  const size_t need = (size_t)nb * (size_t)width * sizeof(float);  // == dh_colsum_f32_workspace_bytes(n_rows, width)
  if (!workspace || workspace_bytes < need) return dh::fail(DH_ERR_WORKSPACE, "dh_colsum_bf16: workspace %zu < %zu bytes", workspace_bytes, need);
  float* partial = nb == 1 ? out : static_cast<float*>(workspace);  // one row block: its sums are the result (no second launch)
  if (width % 8 == 0 && ldx % 8 == 0 && dh::aligned16(X))
    hipLaunchKernelGGL(colsum_bf16_partial_vec_kernel, dim3((unsigned)nb, (unsigned)dh::ceil_div(width, 64)), dim3(256), 0, st, n_rows, width, X, ldx, kColsumRows, partial);
  else if (width > 32) hipLaunchKernelGGL(colsum_bf16_partial_kernel<64>, dim3((unsigned)nb, (unsigned)dh::ceil_div(width, 64)), dim3(256), 0, st, n_rows, width, X, ldx, kColsumRows, partial);
  else hipLaunchKernelGGL(colsum_bf16_partial_kernel<16>, dim3((unsigned)nb, (unsigned)dh::ceil_div(width, 16)), dim3(256), 0, st, n_rows, width, X, ldx, kColsumRows, partial);
  if (nb > 1) hipLaunchKernelGGL(colsum_bf16_final_kernel, dim3((unsigned)dh::ceil_div(width, 256)), dim3(256), 0, st, nb, width, partial, out);
  return dh::check_launch("dh_colsum_bf16");
}
