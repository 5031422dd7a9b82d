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
// Round 3 evaluated every element with three float64 lgamma and (backward) two digamma calls: 151 ms forward + backward at 1M x 2000,
// 0.07 of the 14 ms the 88 GB of traffic need.  Round 4:
//   * counts are integers: Gamma(x + d) / Gamma(d) = prod_{k < x} (d + k), so
//         t1 = log( x! / prod_{k < x} (d + k) ),    digamma(d) - digamma(x + d) = - P'(d) / P(d),  P(d) = prod_{k < x} (d + k)
//     — x multiplications (and the derivative by the product rule in the same loop); the integer power x log(d / m) and log(1 - pi)
//     ride the same product, so the whole negative-binomial branch costs TWO float64 logarithms (that product, in chunks of 16
//     factors so that it stays inside the double range, and log1p(m / d)).  Non-integer or large (> 256) counts take the lgamma /
//     digamma formulas as before.
//   * the x = 0 branch (the majority of a count matrix) needs no gamma function at all and only contributes log(pi + (1 - pi) zn):
//     it runs in fp32 throughout (zero_terms below).  Worst-case ~5e-7 relative per element.
//   * the two branches run in two PHASES per row (zinb_kernel below) instead of side by side in a divergent wavefront: 151 -> 56 ms
//     came from the arithmetic above, the rest from no longer executing the count branch on every 64-gene step.
//   Sums stay float64.  The loss equals the float64 formula to ~1e-7 relative, the gradients to a few 1e-7 of their max-norm.
#include "common.h"

namespace {

constexpr double kEps = 1e-10;

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
  const double t = (m - 1.0) / (m + 1.0);
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
  return r + log(x) - 0.5 / x + t;
}

struct Terms {
  double loss, d_m, d_d, d_p;  // d loss / d (scaled mean, disp, pi)
};

// lg = lgamma(de) + lgamma(x + 1) - lgamma(x + de) + x log(de / me) - log(q)   (me = m + eps, q = 1 - p + eps)
// dg = digamma(de) - digamma(x + de)
// For an integer count the whole of lg is ONE logarithm per 16 factors:  lg = log( prod_{k < x} (k + 1) de / ((de + k) me) / q ).
template <bool GRAD>
__device__ __forceinline__ void gamma_terms(double x, double de, double me, double q, double& lg, double& dg) {
  const int xi = (int)x;
  if ((double)xi == x && xi <= 256) {  // integer count: finite products (16 factors stay inside the double range for every clamp bound)
    lg = 0.0;
    dg = 0.0;
    double extra = 1.0 / q;
    for (int k0 = 0; k0 < xi; k0 += 16) {
      const int k1 = min(xi, k0 + 16);
      double num = 1.0, den = 1.0, pd = 1.0, dpd = 0.0;  // pd = prod (de + k), dpd = d pd / d de
      for (int k = k0; k < k1; ++k) {
        const double f = de + (double)k;
        if (GRAD) dpd = dpd * f + pd;
        pd *= f;
        num *= (double)(k + 1) * de;
        den *= me;
      }
      lg += fast_log(num * extra / (den * pd));
      extra = 1.0;
      if (GRAD) dg -= dpd / pd;
    }
    return;
  }
  lg = lgamma(de) + lgamma(x + 1.0) - lgamma(x + de) + x * log(de / me) - log(q);
  if (GRAD) dg = digamma_pos(de) - digamma_pos(x + de);
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
  const float rs = 1.f / s;
  const float u = (m + 1e-10f) * rs;
  const float lr = u < 0.5f ? log1pf(-u) : logf(d * rs);
  const float zn = expf(d * lr);
  const float w = p + (1.f - p) * zn + 1e-10f;
  o.loss = -logf(w);
  if (GRAD) {
    const float rw = 1.f / w;
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

// x > 0: float64 (the Gamma-function ratios as products, one logarithm per 16 factors)
template <bool GRAD>
__device__ __forceinline__ Terms count_terms(double x, double m, double d, double p, double ridge) {
  Terms o{0.0, 0.0, 0.0, 0.0};
  const double de = d + kEps, me = m + kEps, q = 1.0 - p + kEps;
  double lg, dg;
  gamma_terms<GRAD>(x, de, me, q, lg, dg);
  const double l1 = fast_log1p(m / de);  // log(1 + m / (d + eps))
  o.loss = lg + (d + x) * l1;
  if (GRAD) {
    o.d_p = 1.0 / q;
    o.d_m = (d + x) / (de + m) - x / me;
    o.d_d = dg + l1 - (d + x) * m / (de * (de + m)) + x / de;
  }
  if (ridge > 0.0) {
    o.loss += ridge * p * p;
    if (GRAD) o.d_p += 2.0 * ridge * p;
  }
  return o;
}

// One wavefront per row, two phases.  Phase 1 walks the row 64 genes at a time: the x = 0 elements are finished on the spot (fp32),
// the positions of the x > 0 elements are appended to a per-wave list in LDS (ballot + prefix count: deterministic order).  Phase 2
// evaluates the float64 count terms on the list, 64 at a time with every lane busy — inside phase 1 the one lane in ten with a count
// would make the whole wavefront execute the product loops and the float64 logarithm on every step (that divergence, not the
// arithmetic of the zeros, was why round 3's 151 ms did not depend on the density).  A row with more counts than the list holds
// finishes the overflow inside phase 1.
constexpr int NZ_CAP = 2048;  // list entries per wave (8 KB; 4 waves per block)

template <bool GRAD>
__global__ __launch_bounds__(256) void zinb_kernel(int64_t n, int64_t g, const float* __restrict__ X, int64_t ldx, const float* __restrict__ M,
                                                   int64_t ldm, const float* __restrict__ D, int64_t ldd, const float* __restrict__ P, int64_t ldp,
                                                   const double* __restrict__ sf, double ridge, double* __restrict__ rowloss,
                                                   const double* __restrict__ upstream, float* __restrict__ dM, float* __restrict__ dD,
                                                   float* __restrict__ dP, int64_t ldo) {
  __shared__ int nz_list[4][NZ_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= n) return;
  const double s = sf ? sf[row] : 1.0;
  const double up = GRAD ? upstream[0] : 0.0;  // d(result) / d(element loss) = grad_output / (n g), a device scalar: no host round trip
  const float upf = (float)up, upsf = (float)(up * s), ridge_f = (float)ridge;
  const float* x_row = X + row * ldx;
  const float* m_row = M + row * ldm;
  const float* d_row = D + row * ldd;
  const float* p_row = P + row * ldp;
  int* list = nz_list[wave];
  double acc = 0.0;
  float acc_z = 0.f;  // the zeros of one 64-gene step are summed in fp32 lanes and folded into the float64 sum every 16 steps
  int cnt = 0, folded = 0;
  auto count_element = [&](int64_t c) __attribute__((always_inline)) {
    const Terms t = count_terms<GRAD>((double)x_row[c], (double)m_row[c] * s, (double)d_row[c], (double)p_row[c], ridge);
    if (GRAD) {
      dM[row * ldo + c] = (float)(up * t.d_m * s);
      dD[row * ldo + c] = (float)(up * t.d_d);
      dP[row * ldo + c] = (float)(up * t.d_p);
    } else {
      acc += t.loss;
    }
  };
  for (int64_t c0 = 0; c0 < g; c0 += 64) {
    const int64_t c = c0 + lane;
    const bool in = c < g;
    const float x = in ? x_row[c] : 0.f;
    const bool nz = in && x > 1e-8f;
    if (in && !nz) {
      const ZTerms t = zero_terms<GRAD>((float)((double)m_row[c] * s), d_row[c], p_row[c], ridge_f);
      if (GRAD) {
        dM[row * ldo + c] = upsf * t.d_m;
        dD[row * ldo + c] = upf * t.d_d;
        dP[row * ldo + c] = upf * t.d_p;
      } else {
        acc_z += t.loss;
      }
    }
    const unsigned long long mask = __ballot(nz);
    if (mask) {
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (nz) {
        if (pos < NZ_CAP) list[pos] = (int)c;
        else count_element(c);
      }
      cnt += __popcll(mask);
    }
    if (!GRAD && ++folded == 16) {
      acc += (double)acc_z;
      acc_z = 0.f;
      folded = 0;
    }
  }
  if (!GRAD) acc += (double)acc_z;
  const int listed = cnt < NZ_CAP ? cnt : NZ_CAP;
  for (int i = lane; i < listed; i += 64) count_element(list[i]);  // (a wave reads its own LDS writes: no barrier needed)
  if (!GRAD) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) rowloss[row] = acc;
  }
}

int check(const char* me, int64_t n, int64_t g, const void* X, int64_t ldx, const void* M, int64_t ldm, const void* D, int64_t ldd, const void* P,
          int64_t ldp) {
  if (n < 0 || g < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n == 0 || g == 0) return 1;
  if (!X || !M || !D || !P) return dh::fail(DH_ERR_INVALID, "%s: null matrix", me);
  if (ldx < g || ldm < g || ldd < g || ldp < g) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < n_genes", me);
  return DH_OK;
}

}  // namespace

extern "C" int dh_zinb_nll_forward_f32(int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean, int64_t ldm, const float* disp,
                                       int64_t ldd, const float* pi, int64_t ldp, const double* scale_factor, double ridge_lambda,
                                       double* rowloss, dh_stream_t stream) {
  const int rc = check("dh_zinb_nll_forward_f32", n, g, X, ldx, mean, ldm, disp, ldd, pi, ldp);
  if (rc != DH_OK) return rc > 0 ? DH_OK : rc;
  if (!rowloss) return dh::fail(DH_ERR_INVALID, "dh_zinb_nll_forward_f32: null output");
  hipLaunchKernelGGL(zinb_kernel<false>, dim3((unsigned)dh::ceil_div(n, 4)), dim3(256), 0, dh::as_stream(stream), n, g, X, ldx, mean, ldm, disp, ldd,
                     pi, ldp, scale_factor, ridge_lambda, rowloss, (const double*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0);
  return dh::check_launch("dh_zinb_nll_forward_f32");
}

extern "C" int dh_zinb_nll_backward_f32(int64_t n, int64_t g, const float* X, int64_t ldx, const float* mean, int64_t ldm, const float* disp,
                                        int64_t ldd, const float* pi, int64_t ldp, const double* scale_factor, double ridge_lambda,
                                        const double* upstream, float* d_mean, float* d_disp, float* d_pi, int64_t ldo, dh_stream_t stream) {
  const int rc = check("dh_zinb_nll_backward_f32", n, g, X, ldx, mean, ldm, disp, ldd, pi, ldp);
  if (rc != DH_OK) return rc > 0 ? DH_OK : rc;
  if (!upstream || !d_mean || !d_disp || !d_pi || ldo < g) return dh::fail(DH_ERR_INVALID, "dh_zinb_nll_backward_f32: bad output / upstream");
  hipLaunchKernelGGL(zinb_kernel<true>, dim3((unsigned)dh::ceil_div(n, 4)), dim3(256), 0, dh::as_stream(stream), n, g, X, ldx, mean, ldm, disp, ldd,
                     pi, ldp, scale_factor, ridge_lambda, (double*)nullptr, upstream, d_mean, d_disp, d_pi, ldo);
  return dh::check_launch("dh_zinb_nll_backward_f32");
}
