// One Adam step over a handful of parameter tensors in TWO launches (torch.optim.Adam, amsgrad = False; dance's models all train with it:
// scdeepsort.py:160, graphsc.py:180).  Inside a captured mini-batch step the optimiser was ~14 multi-tensor launches (foreach) or one
// 38 us launch (fused, capturable: the step counters are device tensors and the bias corrections are evaluated per element); here the
// counters are ticked by a one-wave kernel and the update kernel reads them once per workgroup: ~5 us + ~5 us.
//
// Arithmetic, per element, in the order of torch's single-tensor implementation (torch/optim/adam.py _single_tensor_adam):
//   g' = g + weight_decay * p;  m = m + (1 - beta1) (g' - m)  [lerp];  v = beta2 v + (1 - beta2) g' g';
//   p = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// every operation separately rounded in fp32 (-ffp-contract=off), the bias corrections evaluated in double and rounded once.
#include "common.h"

namespace {

constexpr int DH_ADAM_MAX = 8;

struct AdamArgs {
  float* p[DH_ADAM_MAX];
  const float* g[DH_ADAM_MAX];
  float* m[DH_ADAM_MAX];
  float* v[DH_ADAM_MAX];
  float* step[DH_ADAM_MAX];
  int64_t numel[DH_ADAM_MAX];
  int n;
};

__global__ __launch_bounds__(64) void adam_tick_kernel(AdamArgs a) {
  const int i = threadIdx.x;
  if (i < a.n) *a.step[i] += 1.f;
}

__global__ __launch_bounds__(256) void adam_update_kernel(AdamArgs a, float lr, float beta1, float beta2, float eps, float weight_decay) {
  const int t = blockIdx.y;
  const int64_t n = a.numel[t];
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  const double step = (double)*a.step[t];
  const float bc1 = (float)(1.0 - pow((double)beta1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  const float step_size = lr / bc1;
  const float one_minus_b1 = 1.f - beta1, one_minus_b2 = 1.f - beta2;
  float* p = a.p[t];
  const float* g = a.g[t];
  float* m = a.m[t];
  float* v = a.v[t];
  for (int64_t i = i0; i < min(i0 + 4, n); ++i) {
    float gi = g[i];
    if (weight_decay != 0.f) gi = gi + weight_decay * p[i];
    const float mi = m[i] + one_minus_b1 * (gi - m[i]);
    const float vi = v[i] * beta2 + one_minus_b2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

}  // namespace

extern "C" int dh_adam_step_f32(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                float* const* step, const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                                dh_stream_t stream) {
  const char* me = "dh_adam_step_f32";
  if (n < 0) return dh::fail(DH_ERR_INVALID, "%s: negative count", me);
  if (n == 0) return DH_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step || !numel) return dh::fail(DH_ERR_INVALID, "%s: null table", me);
  hipStream_t st = dh::as_stream(stream);
  for (int base = 0; base < n; base += DH_ADAM_MAX) {
    AdamArgs a{};
    a.n = n - base < DH_ADAM_MAX ? n - base : DH_ADAM_MAX;
    int64_t longest = 0;
    for (int i = 0; i < a.n; ++i) {
      const int k = base + i;
      if (numel[k] < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
      if (numel[k] > 0 && (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k])) return dh::fail(DH_ERR_INVALID, "%s: null tensor %d", me, k);
      if (!step[k]) return dh::fail(DH_ERR_INVALID, "%s: null step counter %d", me, k);
      a.p[i] = params[k]; a.g[i] = grads[k]; a.m[i] = exp_avg[k]; a.v[i] = exp_avg_sq[k]; a.step[i] = step[k]; a.numel[i] = numel[k];
      if (numel[k] > longest) longest = numel[k];
    }
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, a);
    if (longest > 0)
      hipLaunchKernelGGL(adam_update_kernel, dim3((unsigned)dh::ceil_div(longest, 1024), (unsigned)a.n), dim3(256), 0, st, a, lr, beta1, beta2, eps,
                         weight_decay);
  }
  return dh::check_launch(me);
}
