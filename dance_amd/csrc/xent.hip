// Cross-entropy with reduction = "sum" over rows of logits — `nn.CrossEntropyLoss(reduction="sum")` of scDeepSort's training step
// (scdeepsort.py:185, :242) — and its gradient, in one pass:
//     lse_i = log sum_j exp(x_ij),   loss = sum_i (lse_i - x_{i, y_i}),   d_ij = softmax(x_i)_j - [j == y_i]      (rows with y_i == ignore: 0)
// torch evaluates it as log_softmax + nll_loss forward, and nll_loss backward + log_softmax backward; its nll_loss reductions run in ONE
// workgroup (55 + 42 us per call at 65536 x 16 — 1.27 of the 17.5 ms of kernels of a 1M-cell scDeepSort epoch, next to 0.13 for the two
// softmax kernels: profiles/r05q_scdeepsort_epoch_kernels_1M_before_xent.md).  Here G lanes share a row (G = the power of two >= the class count, at
// most 64; wider rows are walked in strides of 64), the row is read once into registers when it fits (<= 4 values per lane) and twice
// otherwise, a block's row losses are summed by a fixed tree into partial[block], and a one-block kernel adds the partials in order:
// deterministic, no atomics.  The backward of the autograd function is d times the upstream scalar.
#include "common.h"

namespace {

template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, G));
  return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}

constexpr int XE_REG = 4;  // values of a row a lane keeps in registers (rows up to 4 G classes are read once)

template <int G>
__global__ __launch_bounds__(256) void xent_kernel(int64_t n, int c, const float* __restrict__ X, int64_t ldx, const int64_t* __restrict__ Y,
                                                   int64_t ignore_index, float* __restrict__ D, int64_t ldd, float* __restrict__ partial) {
  constexpr int RPB = 256 / G;
  __shared__ float red[RPB];
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  const int64_t row = (int64_t)blockIdx.x * RPB + rl;
  const bool in = row < n;
  const int64_t rc = in ? row : n - 1;          // clamped: every lane runs the same instruction stream (no load behind a branch)
  const float* x = X + rc * ldx;
  const int64_t y = Y[rc];
  const bool live = in && y != ignore_index;
  const bool fits = c <= XE_REG * G;            // uniform
  float v[XE_REG];
  float m = -INFINITY;
  if (fits) {
#pragma unroll
    for (int k = 0; k < XE_REG; ++k) {
      const int j = g + k * G;
      const float t = x[min(j, c - 1)];
      v[k] = j < c ? t : -INFINITY;
      m = fmaxf(m, v[k]);
    }
  } else {
    for (int j = g; j < c; j += G) m = fmaxf(m, x[j]);
  }
  m = group_max<G>(m);
  float s = 0.f;
  if (fits) {
#pragma unroll
    for (int k = 0; k < XE_REG; ++k) {
      v[k] = __expf(v[k] - m);                  // exp(-inf) = 0 for the padding
      s += v[k];
    }
  } else {
    for (int j = g; j < c; j += G) s += __expf(x[j] - m);
  }
  s = group_sum<G>(s);
  const float rs = 1.f / s;
  if (D) {
    float* d = D + rc * ldd;
    if (fits) {
#pragma unroll
      for (int k = 0; k < XE_REG; ++k) {
        const int j = g + k * G;
        if (in && j < c) d[j] = live ? v[k] * rs - (j == y ? 1.f : 0.f) : 0.f;
      }
    } else if (in) {
      for (int j = g; j < c; j += G) d[j] = live ? __expf(x[j] - m) * rs - (j == y ? 1.f : 0.f) : 0.f;
    }
  }
  // the row's loss: m + log s - x[y].  A label outside [0, c) that is not ignore_index is the caller's error (torch raises a device
  // assertion for it); here it is clamped into range rather than read out of bounds, and its row contributes like class 0 / c - 1
  float li = 0.f;
  if (g == 0 && live) li = m + __logf(s) - x[min(max(y, (int64_t)0), (int64_t)c - 1)];
  if (g == 0) red[rl] = li;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < RPB; ++k) t += red[k];
    partial[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void xent_final_kernel(int64_t n_blocks, const float* __restrict__ partial, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < n_blocks; b += 256) s += (double)partial[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)red[0];
}

int group_of(int64_t c) { return c <= 4 ? 4 : c <= 8 ? 8 : c <= 16 ? 16 : c <= 32 ? 32 : 64; }

}  // namespace

extern "C" size_t dh_softmax_xent_sum_workspace_bytes(int64_t n, int64_t n_classes) {
  if (n <= 0 || n_classes <= 0) return 0;
  return (size_t)dh::ceil_div(n, 256 / group_of(n_classes)) * sizeof(float);
}

extern "C" int dh_softmax_xent_sum_f32(int64_t n, int64_t n_classes, const float* logits, int64_t ldx, const int64_t* labels,
                                       int64_t ignore_index, float* loss, float* d_logits, int64_t ldd, void* workspace,
                                       size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_softmax_xent_sum_f32";
  if (n < 0 || n_classes <= 0 || n_classes > (int64_t)1 << 20) return dh::fail(DH_ERR_INVALID, "%s: bad size", me);
  if (!loss) return dh::fail(DH_ERR_INVALID, "%s: null loss", me);
  hipStream_t st = dh::as_stream(stream);
  if (n == 0) {
    if (dh::zero_async(loss, sizeof(float), st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: clearing the result failed", me);
    return DH_OK;
  }
  if (!logits || !labels || ldx < n_classes || (d_logits && ldd < n_classes)) return dh::fail(DH_ERR_INVALID, "%s: bad operand", me);
  const size_t need = dh_softmax_xent_sum_workspace_bytes(n, n_classes);
  if (!workspace || workspace_bytes < need) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, workspace_bytes, need);
  const int G = group_of(n_classes);
  const int64_t blocks = dh::ceil_div(n, 256 / G);
  float* partial = static_cast<float*>(workspace);
#define DH_XE(GV) \
  hipLaunchKernelGGL(xent_kernel<GV>, dim3((unsigned)blocks), dim3(256), 0, st, n, (int)n_classes, logits, ldx, labels, ignore_index, d_logits, ldd, partial)
  switch (G) {
    case 4: DH_XE(4); break;
    case 8: DH_XE(8); break;
    case 16: DH_XE(16); break;
    case 32: DH_XE(32); break;
    default: DH_XE(64); break;
  }
#undef DH_XE
  int rc = dh::check_launch(me);
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(xent_final_kernel, dim3(1), dim3(256), 0, st, blocks, partial, loss);
  return dh::check_launch("dh_softmax_xent_sum_f32(final)");
}
