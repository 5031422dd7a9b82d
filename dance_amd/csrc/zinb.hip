// Fused zero-inflated negative-binomial NLL — ``ZINBLoss.forward`` of dance/utils/loss.py:780-829 (used by scTAG sctag.py:254,347,
// scDSC scdsc.py:279-283 and scHeteroNet scheteronet.py:289-336, :684-690) and its gradient.
//
// The reference evaluates ~25 elementwise torch ops over the N x G matrices (mean, disp, pi from the decoder heads, the raw counts)
// — each a full pass over HBM, most of them in float64 because the size factors arrive as a float64 tensor and promote the whole
// expression (`mean * scale_factor[:, None]`), and t1 is float64 explicitly (`disp.double()`).  Here one kernel reads the four
// fp32 matrices once and reduces the per-element loss to a float64 sum per row (16 bytes per element instead of ~400), and one
// kernel recomputes the element terms and writes the three gradients (28 bytes per element).
//
//   m = mean * sf,  eps = 1e-10
//   t1 = lgamma(d + eps) + lgamma(x + 1) - lgamma(x + d + eps)
//   t2 = (d + x) log(1 + m / (d + eps)) + x (log(d + eps) - log(m + eps))
//   nb = t1 + t2 - log(1 - pi + eps);   zn = (d / (d + m + eps))^d;   zc = -log(pi + (1 - pi) zn + eps)
//   loss = (x <= 1e-8 ? zc : nb) + ridge * pi^2,   result = mean over all elements
//
// Round 3 evaluated every element with three float64 lgamma and (backward) two digamma calls: 151 ms forward + backward at 1M x 2000.
// Round 4, 24 ms (9 forward + 15 backward; 19 ms at 10 % density; profiles/r04zd_zinb_time.json), in the order of what each step bought:
//   * counts are integers: Gamma(x + d) / Gamma(d) = prod_{k < x} (d + k), so t1 = log( x! / prod_{k < x} (d + k) ) and the integer
//     power x log(d / m) and log(1 - pi) ride the same product, in chunks of 16 factors so that it stays inside the double range; the
//     gradient's digamma(d) - digamma(x + d) + x / d — two large terms that cancel — is the sum (1 / d) sum_k k / (d + k).  No
//     float64 division (rcp_d), logarithms of the float64 products on the fp32 transcendental unit (log_d).      151 -> 56 ms
//   * the x = 0 branch (the majority of a count matrix) needs no gamma function and runs in fp32 throughout (zero_terms).
//   * the two branches no longer share a divergent wavefront: the one lane in ten with a count made every 64-gene step execute the
//     count branch as well.  Both kernels collect the counts of a window of 256 genes in LDS and evaluate them afterwards, 64 at a time.
//   * registers: the library's lgamma (rare fallback) alone held the forward kernel at 204 VGPRs = 2 resident waves per SIMD, and a
//     wave waited out one HBM round trip per 64 genes: own lgamma (92 VGPRs, 5 waves) and the loads of four steps issued together.
//                                                                                                               forward 29 -> 11 ms
//   * backward writes whole lines: windows of 256 genes, gradients staged in LDS (see zinb_backward_kernel).    backward 26 -> 15 ms
//   * forward in the same windows, the counts' operands kept in LDS instead of re-read.                          forward 11 -> 9 ms
//   Sums stay float64.  The loss equals the float64 formula to ~1e-8 relative, the gradients to ~2e-7 of their max-norm.
#include "common.h"

namespace {

constexpr double kEps = 1e-10;

// The transcendental unit's own instructions (v_rcp_f32, v_exp_f32, v_log_f32: 1 ulp) without the library's wrappers: __frcp_rn is a
// correctly rounded division (v_div_scale / v_div_fmas / v_div_fixup and four FMAs: ten instructions), __expf / __logf add a range
// test, two selects and an ldexp for denormal arguments and results.  None of that is needed here — every argument is a normal number
// after the activations' clamps and a result below 2^-126 is clamped or added to something O(1e-10) — and these wrappers were 110 of the
// 215 vector instructions per x = 0 element (ISA count of round 6; the x = 0 path is what the kernels' time is).
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float log_f(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

// 1 / a for a double inside the fp32 range (every argument here is, after the activations' clamps): the fp32 reciprocal and one Newton
// step in float64 — ~1e-14 relative for 3 instructions instead of the ~30 of a float64 division
__device__ __forceinline__ double rcp_d(double a) {
  const double r = (double)rcp_f((float)a);
  return r * (2.0 - a * r);
}
// log(x) for a positive normal double to ~2e-14 relative: x = 2^e m with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(t), t = (m - 1) / (m + 1),
// |t| <= 0.1716: the odd series through t^15 (truncation 2 t^17 / 17 < 2e-14), one division — about a third of the library call,
// which is what the kernels spend their time in once the gamma functions are gone.
__device__ __forceinline__ double fast_log(double x) {
  unsigned long long b = __double_as_longlong(x);
  int e = (int)(b >> 52) - 1023;
  b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m = __longlong_as_double(b);
  if (m > 1.4142135623730951) {
    m *= 0.5;
    e += 1;
  }
  const double t = (m - 1.0) * rcp_d(m + 1.0);  // m + 1 in [1.7, 2.42): the Newton-refined fp32 reciprocal (1e-14) instead of a ~30-instruction division
  const double t2 = t * t;
  double p = 1.0 / 15.0;
  p = p * t2 + 1.0 / 13.0;
  p = p * t2 + 1.0 / 11.0;
  p = p * t2 + 1.0 / 9.0;
  p = p * t2 + 1.0 / 7.0;
  p = p * t2 + 1.0 / 5.0;
  p = p * t2 + 1.0 / 3.0;
  p = p * t2 + 1.0;
  return 2.0 * t * p + (double)e * 0.6931471805599453;
}
// log(1 + x), x >= 0
__device__ __forceinline__ double fast_log1p(double x) {
  if (x < 1e-5) return x * (1.0 - x * (0.5 - x * (1.0 / 3.0)));  // the next term, x^4 / 4, is < 3e-16 relative
  return fast_log(1.0 + x);  // rounding 1 + x costs 1e-16 / x < 1e-11 relative
}

// digamma(x), x > 0: upward recurrence to x >= 6, then the asymptotic series (error < 1e-13 there)
__device__ __forceinline__ double digamma_pos(double x) {
  double r = 0.0;
  while (x < 6.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double f = 1.0 / (x * x);
  const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0)))));
  return r + fast_log(x) - 0.5 / x + t;
}

// log of a positive double whose VALUE may lie far outside the fp32 range (products of up to 16 factors): exponent and mantissa apart,
// the mantissa's logarithm on the fp32 transcendental unit (v_log_f32, ~1 ulp): absolute error ~1e-7, which is what a loss term needs
__device__ __forceinline__ float log_d(double v) {
  int e;
  const double m = frexp(v, &e);  // m in [0.5, 1)
  return ((float)e + __builtin_amdgcn_logf((float)m)) * 0.69314718055994531f;
}
// log(1 + x) in fp32 for x > -1: Kahan's form x log(t) / (t - 1), t = fl(1 + x) — exact in the limit x -> 0 without a series
__device__ __forceinline__ float log1p_f(float x) {
  const float t = 1.f + x;
  const float d = t - 1.f;
  return d == 0.f ? x : log_f(t) * (x * rcp_f(d));  // branch-free: a select on the result
}

// The decoder heads' activations (scdsc.py:601-618, sctag.py:531-548: MeanAct = clamp(exp(a), 1e-5, 1e6), DispAct = clamp(softplus(a),
// 1e-4, 1e4), pi = sigmoid(a)) applied to the raw head outputs inside the loss kernels (the *_logits entry points): as torch ops they
// are 5 forward and ~12 backward elementwise passes over N x G matrices (exp, softplus, sigmoid, two clamps; their backward: two
// compare + where + logical_and per clamp, mul, softplus_backward, sigmoid_backward) — 4.6 of the 27 ms of a scDSC epoch at 100k cells,
// next to 1.9 for the two loss kernels themselves.  j* = d(activated) / d(raw), torch's own backward formulas (clamp: inclusive bounds;
// softplus: threshold 20).  exp on the transcendental unit (v_exp_f32): relative error |a| 2^-24 <= 8e-7 inside the clamp range.
struct HeadActs {
  float m, d, p, jm, jd, jp;
};
template <bool GRAD>
__device__ __forceinline__ HeadActs head_acts(float am, float ad, float ap) {
  HeadActs o{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float e = exp_f(am);
  o.m = fminf(fmaxf(e, 1e-5f), 1e6f);
  const float z = exp_f(ad);
  const float sp = ad > 20.f ? ad : log1p_f(z);
  o.d = fminf(fmaxf(sp, 1e-4f), 1e4f);
  o.p = rcp_f(1.f + exp_f(-ap));
  if (GRAD) {
    o.jm = (e >= 1e-5f && e <= 1e6f) ? e : 0.f;
    const float sg = ad > 20.f ? 1.f : z * rcp_f(z + 1.f);
    o.jd = (sp >= 1e-4f && sp <= 1e4f) ? sg : 0.f;
    o.jp = o.p * (1.f - o.p);
  }
  return o;
}

struct Terms {
  double loss, d_m, d_d, d_p;  // d loss / d (scaled mean, disp, pi)
};

// lgamma(x), x > 0: upward recurrence to x >= 8 (the shifted-out factors collected in one product, one logarithm), then Stirling's
// series through x^-7 (next term 1 / (1188 x^9) < 7e-12 there).  Own code instead of the library's: the fallback below is rare, and
// the library call alone doubled the kernel's register count (4 -> 2 resident waves per SIMD for every element, not just these).
__device__ __forceinline__ double lgamma_pos(double x) {
  double prod = 1.0;
  while (x < 8.0) {
    prod *= x;
    x += 1.0;
  }
  const double r = 1.0 / x, r2 = r * r;
  const double series = r * (1.0 / 12.0 + r2 * (-1.0 / 360.0 + r2 * (1.0 / 1260.0 + r2 * (-1.0 / 1680.0))));
  return (x - 0.5) * fast_log(x) - x + 0.91893853320467274 + series - fast_log(prod);
}

// The general (non-integer or > 256) count:  lg = lgamma(de) + lgamma(x + 1) - lgamma(x + de) + x log(de / me) - log(q),
// dg = digamma(de) - digamma(x + de)   (me = m + eps, q = 1 - p + eps)
template <bool GRAD>
__device__ __forceinline__ void gamma_terms(double x, double de, double me, double q, double& lg, double& dg) {
  lg = lgamma_pos(de) + lgamma_pos(x + 1.0) - lgamma_pos(x + de) + x * fast_log(de / me) - fast_log(q);
  dg = GRAD ? digamma_pos(de) - digamma_pos(x + de) : 0.0;
}

// x = 0 (nine elements in ten of an expression matrix): -log(p + (1 - p) r^d + eps) with r = d / (d + m + eps), all in fp32 — the result
// feeds an fp32 gradient or a float64 sum of ~1e9 terms; the arguments are O(1) after the activations' clamps (mean >= 1e-5, disp in
// [1e-4, 1e4]) and log r goes through log1p(-u) for small u = m / s, so d * log r loses nothing to cancellation.
struct ZTerms {
  float loss, d_m, d_d, d_p;
};
template <bool GRAD>
__device__ __forceinline__ ZTerms zero_terms(float m, float d, float p, float ridge) {
  ZTerms o{0.f, 0.f, 0.f, 0.f};
  const float s = d + m + 1e-10f;
  const float rs = rcp_f(s);
  const float u = (m + 1e-10f) * rs;
  // log r = log1p(-u) (Kahan's form) for u < 1/2, log(d / s) otherwise: ONE logarithm of the selected argument and one reciprocal,
  // no branch (the two-sided form compiled to a divergent branch around two logarithms)
  const bool small = u < 0.5f;
  const float t = 1.f - u, dt = t - 1.f;
  const float lg = log_f(small ? t : d * rs);
  const float lr = small ? (dt == 0.f ? -u : lg * (-u * rcp_f(dt))) : lg;
  const float zn = exp_f(d * lr);
  const float w = p + (1.f - p) * zn + 1e-10f;
  o.loss = -log_f(w);
  if (GRAD) {
    const float rw = rcp_f(w);
    const float dzc_dzn = -(1.f - p) * rw;
    o.d_p = -(1.f - zn) * rw;
    o.d_m = dzc_dzn * (-zn * d * rs);
    o.d_d = dzc_dzn * zn * (lr + u);
  }
  if (ridge > 0.f) {
    o.loss += ridge * p * p;
    if (GRAD) o.d_p += 2.f * ridge * p;
  }
  return o;
}

// x > 0.  Integer counts (what a count matrix holds): the Gamma-function ratios are products,
//     lgamma(de) + lgamma(x + 1) - lgamma(x + de) + x log(de / me) - log q = sum over chunks of 16 factors of log(num) - log(den),
// float64 products, their logarithms by log_d; and in the gradient digamma(de) - digamma(x + de) + x / de — two large terms that cancel
// — is the cancellation-free sum (1 / de) sum_{k < x} k / (de + k).  No float64 division anywhere (rcp_d).  Anything else (a
// non-integer "count", x > 256) takes the lgamma / digamma formulas.
template <bool GRAD>
__device__ __forceinline__ Terms count_terms(double x, double m, double d, double p, double ridge) {
  Terms o{0.0, 0.0, 0.0, 0.0};
  const double de = d + kEps, me = m + kEps, q = 1.0 - p + kEps;
  const int xi = (int)x;
  // a chunk is a product of 16 factors me (de + k) resp. (k + 1) de: both stay inside float64 only while every factor lies within
  // ~1e-19 .. 1e19.  The activations' clamps (mean >= 1e-5, disp in [1e-4, 1e4]) guarantee that; a caller of the C ABI with
  // unclamped operands (me = de = eps: factors of 1e-20, a chunk underflows to 0 and its logarithm is -inf) takes the lgamma path
  const bool in_range = me * de > 1e-18 && me * (de + 256.0) < 1e18 && de < 1e16;
  if ((double)xi == x && xi <= 256 && in_range) {
    float lg = -log_d(q);
    double ksum = 0.0;  // sum_k k / (de + k)
    for (int k0 = 0; k0 < xi; k0 += 16) {
      const int k1 = min(xi, k0 + 16);
      double num = 1.0, den = 1.0;
      for (int k = k0; k < k1; ++k) {
        const double f = de + (double)k;
        num *= (double)(k + 1) * de;
        den *= me * f;
        if (GRAD) ksum += (double)((float)k * rcp_f((float)f));
      }
      lg += log_d(num) - log_d(den);
    }
    const double rde = rcp_d(de), rdm = rcp_d(de + m);
    const double r = m * rde;
    // log(1 + m / (d + eps)) in float64: it is multiplied by (d + x) in the loss and, in d_d, a nearly equal term is subtracted from it —
    // the fp32 logarithm's 6e-8 absolute error was 6e-4 per element at d = 1e4 and an O(1) relative error of d_d for d >~ 1e3 m
    const double l1 = fast_log1p(r);
    o.loss = (double)lg + (d + x) * l1;
    if (GRAD) {
      o.d_p = rcp_d(q);
      o.d_m = (d + x) * rdm - x * rcp_d(me);
      o.d_d = ksum * rde + l1 - (d + x) * m * rde * rdm;
    }
  } else {
    double lg, dg;
    gamma_terms<GRAD>(x, de, me, q, lg, dg);
    const double l1 = fast_log1p(m / de);
    o.loss = lg + (d + x) * l1;
    if (GRAD) {
      o.d_p = 1.0 / q;
      o.d_m = (d + x) / (de + m) - x / me;
      o.d_d = dg + l1 - (d + x) * m / (de * (de + m)) + x / de;
    }
  }
  if (ridge > 0.0) {
    o.loss += ridge * p * p;
    if (GRAD) o.d_p += 2.0 * ridge * p;
  }
  return o;
}

// Forward: one wavefront per row, two phases per WINDOW of ZU x 64 genes.  Phase 1 finishes the window's x = 0 elements on the spot
// (fp32) and appends the positions (and operands) of its x > 0 elements to a per-wave list in LDS (ballot + prefix count:
// deterministic order); phase 2 evaluates the float64 count terms on that list, 64 at a time with every lane busy — side by side in
// phase 1 the one lane in ten with a count would make the whole wavefront execute the product loops and the float64 logarithm on
// every step (that divergence, not the arithmetic of the zeros, was why round 3's 151 ms did not depend on the density).
constexpr int ZU = 4;          // 64-gene steps whose loads are issued together: a wave keeps 4 x 4 loads per lane in flight

template <bool LOGITS>
__global__ __launch_bounds__(256) void zinb_forward_kernel(int64_t n, int64_t g, const float* __restrict__ X, int64_t ldx,
                                                           const float* __restrict__ M, int64_t ldm, const float* __restrict__ D, int64_t ldd,
                                                           const float* __restrict__ P, int64_t ldp, const double* __restrict__ sf, double ridge,
                                                           double* __restrict__ rowloss) {
  constexpr int W = 64 * ZU;
  __shared__ float stage[4][4][W];  // the window's x, scaled-mean input, disp, pi: the counts are evaluated from here, not re-read
  __shared__ unsigned short wlist[4][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= n) return;
  const double s = sf ? sf[row] : 1.0;
  const float ridge_f = (float)ridge;
  const float* x_row = X + row * ldx;
  const float* m_row = M + row * ldm;
  const float* d_row = D + row * ldd;
  const float* p_row = P + row * ldp;
  float (*st)[W] = stage[wave];
  unsigned short* wl = wlist[wave];
  double acc = 0.0;
  for (int64_t c0 = 0; c0 < g; c0 += W) {
    float xv[ZU], mv[ZU], dv[ZU], pv[ZU];
#pragma unroll
    for (int u = 0; u < ZU; ++u) {  // all loads of the window first (clamped addresses: no branch between them)
      const int64_t c = c0 + 64 * u + lane, cc = c < g ? c : g - 1;
      xv[u] = x_row[cc];
      mv[u] = m_row[cc];
      dv[u] = d_row[cc];
      pv[u] = p_row[cc];
    }
    if (LOGITS) {
#pragma unroll
      for (int u = 0; u < ZU; ++u) {
        const HeadActs a = head_acts<false>(mv[u], dv[u], pv[u]);
        mv[u] = a.m, dv[u] = a.d, pv[u] = a.p;
      }
    }
    int cnt = 0;
    float acc_z = 0.f;  // the window's zeros in fp32 lanes, folded into the float64 sum per window
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int li = 64 * u + lane;
      const bool in = c0 + li < g;
      const bool nz = in && xv[u] > 1e-8f;
      if (in && !nz) acc_z += zero_terms<false>((float)((double)mv[u] * s), dv[u], pv[u], ridge_f).loss;
      const unsigned long long mask = __ballot(nz);
      if (nz) {
        const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        wl[pos] = (unsigned short)li;
        st[0][li] = xv[u];
        st[1][li] = mv[u];
        st[2][li] = dv[u];
        st[3][li] = pv[u];
      }
      cnt += __popcll(mask);
    }
    acc += (double)acc_z;
    for (int i = lane; i < cnt; i += 64) {  // the window's counts, every lane busy (a wave reads its own LDS writes: no barrier)
      const int li = wl[i];
      acc += count_terms<false>((double)st[0][li], (double)st[1][li] * s, (double)st[2][li], (double)st[3][li], ridge).loss;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) rowloss[row] = acc;
}

// Backward: the same split of the two branches, but inside WINDOWS of ZU x 64 genes, and every gradient goes through a per-wave LDS
// stage so that the three outputs are written as whole 256-byte runs.  (With the row-long list of the forward kernel the x = 0 lanes
// wrote their lines with holes and the counts filled them in much later — after the partly written lines had left L2: the 24 GB of
// gradient writes turned into read-modify-writes and the kernel took 23 - 26 ms whatever the arithmetic cost.)
template <bool LOGITS>
__global__ __launch_bounds__(256) void zinb_backward_kernel(int64_t n, int64_t g, const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ M, int64_t ldm, const float* __restrict__ D, int64_t ldd,
                                                            const float* __restrict__ P, int64_t ldp, const double* __restrict__ sf, double ridge,
                                                            const double* __restrict__ upstream, float* __restrict__ dM, float* __restrict__ dD,
                                                            float* __restrict__ dP, int64_t ldo) {
  constexpr int W = 64 * ZU;
  __shared__ float stage[4][3][W];
  __shared__ unsigned short wlist[4][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= n) return;
  const double s = sf ? sf[row] : 1.0;
  const double up = upstream[0];  // d(result) / d(element loss) = grad_output / (n g), a device scalar: no host round trip
  const float upf = (float)up, upsf = (float)(up * s), ridge_f = (float)ridge;
  const float* x_row = X + row * ldx;
  const float* m_row = M + row * ldm;
  const float* d_row = D + row * ldd;
  const float* p_row = P + row * ldp;
  float (*st)[W] = stage[wave];
  unsigned short* wl = wlist[wave];
  for (int64_t c0 = 0; c0 < g; c0 += W) {
    float xv[ZU], mv[ZU], dv[ZU], pv[ZU];
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int64_t c = c0 + 64 * u + lane, cc = c < g ? c : g - 1;
      xv[u] = x_row[cc];
      mv[u] = m_row[cc];
      dv[u] = d_row[cc];
      pv[u] = p_row[cc];
    }
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int li = 64 * u + lane;
      const bool in = c0 + li < g;
      const bool nz = in && xv[u] > 1e-8f;
      if (in && !nz) {
        float jm = 1.f, jd = 1.f, jp = 1.f;
        if (LOGITS) {
          const HeadActs a = head_acts<true>(mv[u], dv[u], pv[u]);
          mv[u] = a.m, dv[u] = a.d, pv[u] = a.p, jm = a.jm, jd = a.jd, jp = a.jp;
        }
        const ZTerms t = zero_terms<true>((float)((double)mv[u] * s), dv[u], pv[u], ridge_f);
        st[0][li] = upsf * t.d_m * jm;
        st[1][li] = upf * t.d_d * jd;
        st[2][li] = upf * t.d_p * jp;
      }
      const unsigned long long mask = __ballot(nz);
      if (nz) wl[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)li;
      cnt += __popcll(mask);
    }
    for (int i = lane; i < cnt; i += 64) {  // the window's counts, every lane busy (a wave reads its own LDS writes: no barrier)
      const int li = wl[i];
      const int64_t c = c0 + li;
      float mc = m_row[c], dc = d_row[c], pc = p_row[c];
      double jm = 1.0, jd = 1.0, jp = 1.0;
      if (LOGITS) {
        const HeadActs a = head_acts<true>(mc, dc, pc);
        mc = a.m, dc = a.d, pc = a.p, jm = (double)a.jm, jd = (double)a.jd, jp = (double)a.jp;
      }
      const Terms t = count_terms<true>((double)x_row[c], (double)mc * s, (double)dc, (double)pc, ridge);
      st[0][li] = (float)(up * t.d_m * s * jm);
      st[1][li] = (float)(up * t.d_d * jd);
      st[2][li] = (float)(up * t.d_p * jp);
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u) {
      const int li = 64 * u + lane;
      const int64_t c = c0 + li;
      if (c < g) {
        dM[row * ldo + c] = st[0][li];
        dD[row * ldo + c] = st[1][li];
        dP[row * ldo + c] = st[2][li];
      }
    }
  }
}

// ---- the three heads' loss, gradients and bias gradients in ONE pass (dh_zinb_heads_fused_f32) -------------------------------------------
// The training loop of scDSC (scdsc.py:265-283) runs forward and backward of the loss on the same operands back to back: the forward
// kernel reads the four N x G matrices (32 GB at 1M x 2000) for a scalar, the backward kernel reads them again, and a third pass
// (dh_colsum_f32, 24 GB) sums the three gradients' columns for the heads' biases: 9.7 + 15.3 + 4.9 ms of a 201 ms epoch.  This kernel
// evaluates each element once: loss (float64 partial per wavefront), the gradients w.r.t. the raw head outputs for a UNIT upstream
// gradient times ``unit`` (= 1 / (N G): the caller folds the real upstream scalar into the heads' small dW / db), written OVER the raw
// outputs (a wavefront owns its elements: read, then write), and the gradients' column sums per block of 64 rows.
//   * a wavefront owns a window of 256 genes x 64 rows: a lane holds 4 ADJACENT genes (one 16-byte load per matrix and row instead
//     of four 4-byte ones: a 4-byte-per-lane load instruction costs the same issue time and moves a quarter), the next row's four
//     loads are in flight while this row is evaluated, the column sums of its 4 genes stay in 12 registers;
//   * x = 0 arithmetic (fp32) runs for every element, branch-free; the counts of the row's window are listed in LDS (ballot order:
//     deterministic) and evaluated afterwards 64 at a time in float64 (the split of the two kernels above), their results merged by a
//     select on the way out;
//   * no barrier, no atomics: wavefronts are independent.
constexpr int FW = 256;        // genes per window
constexpr int FR = 64;         // rows per wavefront

template <bool VEC>
__device__ __forceinline__ void load_quad(float (&v)[4], const float* __restrict__ row, int64_t c, int64_t g) {
  if constexpr (VEC) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 q = *reinterpret_cast<const f32x4*>(row + (c < g ? c : 0));  // g % 4 == 0: a quad is entirely in or out
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = q[j];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = row[c + j < g ? c + j : g - 1];
  }
}
template <bool VEC>
__device__ __forceinline__ void store_quad(float* __restrict__ row, int64_t c, int64_t g, const float (&v)[4]) {
  if constexpr (VEC) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (c < g) *reinterpret_cast<f32x4*>(row + c) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c + j < g) row[c + j] = v[j];
  }
}

// (four waves per SIMD asked for: 128 registers with 18 spilled ones in the rare gamma-function path against 151 and three waves —
// 12.9 -> 12.3 ms at 10 % non-zero, 19.1 -> 18.4 at 57 %, A/B builds on one box)
template <bool VEC>
__global__ __launch_bounds__(256, 4) void zinb_heads_fused_kernel(int64_t n, int64_t g, int nwin, const float* __restrict__ X, int64_t ldx,
                                                               float* __restrict__ M, float* __restrict__ D, float* __restrict__ P, int64_t ld,
                                                               const double* __restrict__ sf, double ridge, double unit,
                                                               double* __restrict__ loss_partials, float* __restrict__ col_partials) {
  __shared__ __attribute__((aligned(16))) float stage_in[4][4][FW];   // per wave: x, raw mean / disp / pi of the row's window
  __shared__ __attribute__((aligned(16))) float stage_out[4][3][FW];  // per wave: the counts' gradients
  __shared__ unsigned short wlist[4][FW];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t unit_id = (int64_t)blockIdx.x * 4 + wave;  // (row block, window)
  const int64_t n_rb = (n + FR - 1) / FR;
  if (unit_id >= n_rb * nwin) return;
  const int64_t rb = unit_id / nwin;
  const int win = (int)(unit_id % nwin);
  const int64_t r0 = rb * FR, r1 = min(n, r0 + FR);
  const int64_t c = (int64_t)win * FW + 4 * lane;  // this lane's first gene
  const float ridge_f = (float)ridge;
  float (*si)[FW] = stage_in[wave];
  float (*so)[FW] = stage_out[wave];
  unsigned short* wl = wlist[wave];
  float colacc[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  double acc = 0.0;

  float xn[4], mn[4], dn[4], pn[4];
  load_quad<VEC>(xn, X + r0 * ldx, c, g);
  load_quad<VEC>(mn, M + r0 * ld, c, g);
  load_quad<VEC>(dn, D + r0 * ld, c, g);
  load_quad<VEC>(pn, P + r0 * ld, c, g);
  for (int64_t row = r0; row < r1; ++row) {
    float xv[4], mv[4], dv[4], pv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = xn[j], mv[j] = mn[j], dv[j] = dn[j], pv[j] = pn[j];
    {
      const int64_t nr = row + 1 < r1 ? row + 1 : row;  // the next row's operands: in flight during this row's arithmetic
      load_quad<VEC>(xn, X + nr * ldx, c, g);
      load_quad<VEC>(mn, M + nr * ld, c, g);
      load_quad<VEC>(dn, D + nr * ld, c, g);
      load_quad<VEC>(pn, P + nr * ld, c, g);
    }
    const double s = sf ? sf[row] : 1.0;
    const float upf = (float)unit, upsf = (float)(unit * s);
    *reinterpret_cast<f32x4*>(&si[0][4 * lane]) = f32x4{xv[0], xv[1], xv[2], xv[3]};
    *reinterpret_cast<f32x4*>(&si[1][4 * lane]) = f32x4{mv[0], mv[1], mv[2], mv[3]};
    *reinterpret_cast<f32x4*>(&si[2][4 * lane]) = f32x4{dv[0], dv[1], dv[2], dv[3]};
    *reinterpret_cast<f32x4*>(&si[3][4 * lane]) = f32x4{pv[0], pv[1], pv[2], pv[3]};
    // x = 0 arithmetic for every element, two at a time (NOT unrolled: with the four elements of a lane interleaved the kernel held
    // 187 registers = 2 resident waves per SIMD; operands come back from the LDS stage so that nothing is indexed by the loop counter
    // in registers): results to the output stage, counts listed
    int cnt = 0;
#pragma nounroll
    for (int jp = 0; jp < 2; ++jp) {
      const int l0 = 4 * lane + 2 * jp;
      const f32x2 x2 = *reinterpret_cast<const f32x2*>(&si[0][l0]);
      const f32x2 m2 = *reinterpret_cast<const f32x2*>(&si[1][l0]);
      const f32x2 d2 = *reinterpret_cast<const f32x2*>(&si[2][l0]);
      const f32x2 p2 = *reinterpret_cast<const f32x2*>(&si[3][l0]);
      f32x2 om, od, op;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool in = c + 2 * jp + e < g;
        const bool nz = in && x2[e] > 1e-8f;
        const HeadActs a = head_acts<true>(m2[e], d2[e], p2[e]);
        const ZTerms t = zero_terms<true>((float)((double)a.m * s), a.d, a.p, ridge_f);
        om[e] = in ? upsf * t.d_m * a.jm : 0.f;
        od[e] = in ? upf * t.d_d * a.jd : 0.f;
        op[e] = in ? upf * t.d_p * a.jp : 0.f;
        acc += (in && !nz) ? (double)t.loss : 0.0;
        const unsigned long long mask = __ballot(nz);
        if (nz) wl[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)(l0 + e);
        cnt += __popcll(mask);
      }
      *reinterpret_cast<f32x2*>(&so[0][l0]) = om;
      *reinterpret_cast<f32x2*>(&so[1][l0]) = od;
      *reinterpret_cast<f32x2*>(&so[2][l0]) = op;
    }
    for (int i = lane; i < cnt; i += 64) {  // the window's counts, every lane busy (a wave reads its own LDS writes: no barrier)
      const int li = wl[i];
      const HeadActs a = head_acts<true>(si[1][li], si[2][li], si[3][li]);
      const Terms t = count_terms<true>((double)si[0][li], (double)a.m * s, (double)a.d, (double)a.p, ridge);
      acc += t.loss;
      so[0][li] = (float)(unit * t.d_m * s * (double)a.jm);
      so[1][li] = (float)(unit * t.d_d * (double)a.jd);
      so[2][li] = (float)(unit * t.d_p * (double)a.jp);
    }
    const f32x4 cm = *reinterpret_cast<const f32x4*>(&so[0][4 * lane]);
    const f32x4 cd = *reinterpret_cast<const f32x4*>(&so[1][4 * lane]);
    const f32x4 cp = *reinterpret_cast<const f32x4*>(&so[2][4 * lane]);
    float gm[4], gd[4], gp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[j] = cm[j], gd[j] = cd[j], gp[j] = cp[j];
      colacc[0][j] += gm[j];
      colacc[1][j] += gd[j];
      colacc[2][j] += gp[j];
    }
    store_quad<VEC>(M + row * ld, c, g, gm);
    store_quad<VEC>(D + row * ld, c, g, gd);
    store_quad<VEC>(P + row * ld, c, g, gp);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) loss_partials[unit_id] = acc;
  float* cpart = col_partials + rb * 3 * g;  // [row block][head][gene]
#pragma unroll
  for (int hd = 0; hd < 3; ++hd)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c + j < g) cpart[(int64_t)hd * g + c + j] = colacc[hd][j];
}

int check(const char* me, int64_t n, int64_t g, const void* X, int64_t ldx, const void* M, int64_t ldm, const void* D, int64_t ldd, const void* P,
          int64_t ldp) {
  if (n < 0 || g < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n == 0 || g == 0) return 1;
  if (!X || !M || !D || !P) return dh::fail(DH_ERR_INVALID, "%s: null matrix", me);
  if (ldx < g || ldm < g || ldd < g || ldp < g) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < n_genes", me);
  return DH_OK;
}

template <bool LOGITS>
int forward_impl(const char* me, int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean, int64_t ldm, const float* disp, int64_t ldd,
                 const float* pi, int64_t ldp, const double* scale_factor, double ridge_lambda, double* rowloss, dh_stream_t stream) {
  const int rc = check(me, n, g, X, ldx, mean, ldm, disp, ldd, pi, ldp);
  if (rc != DH_OK) return rc > 0 ? DH_OK : rc;
  if (!rowloss) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  hipLaunchKernelGGL(zinb_forward_kernel<LOGITS>, dim3((unsigned)dh::ceil_div(n, 4)), dim3(256), 0, dh::as_stream(stream), n, g, X, ldx, mean, ldm,
                     disp, ldd, pi, ldp, scale_factor, ridge_lambda, rowloss);
  return dh::check_launch(me);
}

template <bool LOGITS>
int backward_impl(const char* me, int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean, int64_t ldm, const float* disp, int64_t ldd,
                  const float* pi, int64_t ldp, const double* scale_factor, double ridge_lambda, const double* upstream, float* d_mean,
                  float* d_disp, float* d_pi, int64_t ldo, dh_stream_t stream) {
  const int rc = check(me, n, g, X, ldx, mean, ldm, disp, ldd, pi, ldp);
  if (rc != DH_OK) return rc > 0 ? DH_OK : rc;
  if (!upstream || !d_mean || !d_disp || !d_pi || ldo < g) return dh::fail(DH_ERR_INVALID, "%s: bad output / upstream", me);
  hipLaunchKernelGGL(zinb_backward_kernel<LOGITS>, dim3((unsigned)dh::ceil_div(n, 4)), dim3(256), 0, dh::as_stream(stream), n, g, X, ldx, mean, ldm,
                     disp, ldd, pi, ldp, scale_factor, ridge_lambda, upstream, d_mean, d_disp, d_pi, ldo);
  return dh::check_launch(me);
}

}  // namespace

extern "C" int dh_zinb_nll_forward_f32(int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean, int64_t ldm, const float* disp,
                                       int64_t ldd, const float* pi, int64_t ldp, const double* scale_factor, double ridge_lambda,
                                       double* rowloss, dh_stream_t stream) {
  return forward_impl<false>("dh_zinb_nll_forward_f32", n, g, X, ldx, mean, ldm, disp, ldd, pi, ldp, scale_factor, ridge_lambda, rowloss, stream);
}

extern "C" int dh_zinb_nll_backward_f32(int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean, int64_t ldm, const float* disp,
                                        int64_t ldd, const float* pi, int64_t ldp, const double* scale_factor, double ridge_lambda,
                                        const double* upstream, float* d_mean, float* d_disp, float* d_pi, int64_t ldo, dh_stream_t stream) {
  return backward_impl<false>("dh_zinb_nll_backward_f32", n, g, X, ldx, mean, ldm, disp, ldd, pi, ldp, scale_factor, ridge_lambda, upstream, d_mean,
                              d_disp, d_pi, ldo, stream);
}

// The same loss as a function of the three heads' RAW outputs (the Linear layers' results, before MeanAct / DispAct / Sigmoid):
// the activations and their Jacobians are applied inside the kernels (head_acts above); d_* are the gradients w.r.t. the raw outputs.
extern "C" int dh_zinb_nll_logits_forward_f32(int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean_raw, int64_t ldm,
                                              const float* disp_raw, int64_t ldd, const float* pi_raw, int64_t ldp, const double* scale_factor,
                                              double ridge_lambda, double* rowloss, dh_stream_t stream) {
  return forward_impl<true>("dh_zinb_nll_logits_forward_f32", n, g, X, ldx, mean_raw, ldm, disp_raw, ldd, pi_raw, ldp, scale_factor, ridge_lambda,
                            rowloss, stream);
}

extern "C" int dh_zinb_nll_logits_backward_f32(int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean_raw, int64_t ldm,
                                               const float* disp_raw, int64_t ldd, const float* pi_raw, int64_t ldp, const double* scale_factor,
                                               double ridge_lambda, const double* upstream, float* d_mean_raw, float* d_disp_raw, float* d_pi_raw,
                                               int64_t ldo, dh_stream_t stream) {
  return backward_impl<true>("dh_zinb_nll_logits_backward_f32", n, g, X, ldx, mean_raw, ldm, disp_raw, ldd, pi_raw, ldp, scale_factor, ridge_lambda,
                             upstream, d_mean_raw, d_disp_raw, d_pi_raw, ldo, stream);
}

// Loss, gradients (over the raw outputs, in place) and per-row-block column sums of the three heads in one pass: see
// zinb_heads_fused_kernel.  loss_partials: dh_zinb_heads_fused_partials(n, g, &n_loss, &n_row_blocks) doubles, summed by the caller in
// index order; col_partials: [n_row_blocks][3][g] floats, whose column sums (dh_colsum_f32 over the [n_row_blocks, 3 g] matrix) are the
// three bias gradients for the unit upstream.
extern "C" int dh_zinb_heads_fused_partials(int64_t n, int64_t g, int64_t* n_loss, int64_t* n_row_blocks) {
  if (n < 0 || g < 0 || !n_loss || !n_row_blocks) return dh::fail(DH_ERR_INVALID, "dh_zinb_heads_fused_partials: bad argument");
  const int64_t n_rb = (n + FR - 1) / FR, nwin = (g + FW - 1) / FW;
  *n_loss = n_rb * nwin;
  *n_row_blocks = n_rb;
  return DH_OK;
}

extern "C" int dh_zinb_heads_fused_f32(int64_t n, int64_t g, const float* X, int64_t ldx, float* mean_raw, float* disp_raw, float* pi_raw,
                                       int64_t ld, const double* scale_factor, double ridge_lambda, double unit, double* loss_partials,
                                       float* col_partials, dh_stream_t stream) {
  const char* me = "dh_zinb_heads_fused_f32";
  const int rc = check(me, n, g, X, ldx, mean_raw, ld, disp_raw, ld, pi_raw, ld);
  if (rc != DH_OK) return rc > 0 ? DH_OK : rc;
  if (!loss_partials || !col_partials) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  if (g > 0x7fffffffLL / 2) return dh::fail(DH_ERR_INVALID, "%s: too many genes", me);
  const int64_t n_rb = (n + FR - 1) / FR, nwin = (g + FW - 1) / FW;
  const int64_t blocks = dh::ceil_div(n_rb * nwin, 4);
  if (blocks > 0x7fffffffLL) return dh::fail(DH_ERR_INVALID, "%s: grid too large", me);
  const bool vec = g % 4 == 0 && ldx % 4 == 0 && ld % 4 == 0 && dh::aligned16(X) && dh::aligned16(mean_raw) && dh::aligned16(disp_raw) &&
                   dh::aligned16(pi_raw);
  hipStream_t st = dh::as_stream(stream);
  if (vec)
    hipLaunchKernelGGL(zinb_heads_fused_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, n, g, (int)nwin, X, ldx, mean_raw, disp_raw, pi_raw,
                       ld, scale_factor, ridge_lambda, unit, loss_partials, col_partials);
  else
    hipLaunchKernelGGL(zinb_heads_fused_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, n, g, (int)nwin, X, ldx, mean_raw, disp_raw, pi_raw,
                       ld, scale_factor, ridge_lambda, unit, loss_partials, col_partials);
  return dh::check_launch(me);
}
