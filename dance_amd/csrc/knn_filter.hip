// Matrix-core filter + exact re-rank for the brute-force kNN (dh_knn_bruteforce_f32, algo = DH_KNN_FILTER).
//
// The kNN result is DEFINED by the sequential fp32 distance chain of knn.hip (that is what makes index lists bit-exact
// against the oracle), which costs 3 VALU ops per (pair, feature).  The matrix cores cannot evaluate that chain, but
// they can discard almost every pair first:
//
//   1. tau_q  = exact k-th chain distance of query q inside a strided SAMPLE of S candidates (knn.hip scan).  The true
//               k-th distance over all candidates can only be smaller, so every true neighbour c has d2(q,c) <= tau_q.
//   2. filter = for all pairs, d2~(q,c) = |q|^2 + |c|^2 - 2 q.c on v_mfma_f32_32x32x16_bf16, with every fp32 feature
//               split into two bf16 terms (x = hi + lo + r, |r| <= 2^-16 |x|) and q.c ~ hi.hi + hi.lo + lo.hi (one GEMM
//               with K = 3 d).  |d2~ - d2| <= eps (|q|^2 + |c|^2) with eps = 2^-14 + d 2^-22 (derivation below), so the
//               pairs with d2~ <= tau_q (1 + (d + 16) u) + eps (|q|^2 + |c|^2) are a superset of the true neighbours; they are appended
//               to a per-query survivor list (~ k n / S entries).
//   3. rerank = the exact chain on the survivors only, k smallest (d2, index) per query — the same keys, hence the same
//               bits, as the full scan.  A query whose list overflowed re-scans all candidates inside the same kernel.
//
// Error budget of step 2, three-term form (d > 64; u = 2^-24; all worst case, no statistics), N = |q|^2 + |c|^2 of the centred rows:
//   representation: dropped lo.lo and r terms                     <= 3.1 * 2^-16 sum|q_i c_i|
//   MFMA accumulation over 3d bf16 products (exact each)          <= 3d u (1 + 2^-7)^2 sum|q_i c_i|
//   norms (accumulated in double, one rounding), Cn, the final fma <= 8 u N
//   centring x' = fl(x - mu) moves d2 by                          <= 4 u N
// with sum|q_i c_i| <= N / 2 and the factor 2 in front of the dot product:
//   |d2~ - d2'| <= N (3.1 * 2^-16 + (3.06 d + 12) u)  <  N (2^-14 + d 2^-22) = eps N       (d 2^-22 = 4 d u)
// and the chain itself (the DEFINITION of the distance) differs from the real d2 only relative to d2, not to the norms:
//   d2 <= chain (1 + (d + 2) u (1 + u)^d),  so a true neighbour (chain <= tau) has  d2' <= tau (1 + (d + 16) u) + 4 u N.
// Round 4 tightened this: the old form charged the chain and an fp32 norm sum against the NORMS (eps = 2^-13 + d 2^-20, 2e-3
// at d = 2000) — on clustered 2000-d rows (N = 17x the neighbour distances) that slack alone was two standard deviations of the
// within-cluster distance distribution and every survivor list overflowed (23 s of re-scans at 100k x 2000 instead of 0.4 s).
// The margin scales with the NORMS, so the filter works on CENTRED copies x' = fl(x - mu), mu = column means: distances
// are translation invariant (any mu is admissible — its rounding only costs efficiency), the subtraction's own rounding
// moves d2 by <= 2 |x-y| u (|x'| + |y'|) <= 4 u (|x'|^2 + |y'|^2), well inside the slack above, and embeddings with a
// large common offset (or non-negative expression rows) no longer inflate the survivor lists.  Inputs must be finite.
//
// d <= 64 (what NeighborGraph feeds: PCA embeddings): ONE fp16 term instead of three bf16 terms, thresholds in up to three passes.
//   * The centred data are scaled by a power of two (exact) so that max |y| lies in [2^8, 2^9) and every row becomes ONE
//     fp16 vector yh (11 significant bits, |yh - y| <= 2^-11 |y|, or <= 2^-14 absolute below fp16's normal range — the
//     bound does not rely on subnormal support).  With sum |q_i c_i| <= (|q|^2 + |c|^2) / 2 and the factor 2 of the dot product:
//         |d2~ - d2| <= (|q|^2 + |c|^2) (2^-10 (1 + 2^-12) + (K + 10) 5 u) + 2^-13 sqrt(d) (|q| + |c|) + 1/2
//     (representation; fp32 accumulation over K <= 80 exact products and the threshold columns; flushed elements; the
//     fp16 splits of the two threshold terms), against the test's slack eps (|q|^2 + |c|^2) + A (|q| + |c|) + 1 with
//     eps = 2^-10 + 2^-13 + d 2^-20, A = 1.01 * 2^-13 sqrt(d).  The slack is ~9x the three-term form's, the matrix-core
//     work 2.75x smaller at d = 50 (K = 64 instead of 176); all norms are those of the CENTRED, SCALED rows.
//   * The whole test rides the matrix cores: A row = [-2 yh | -Rq split in 3 | M M M | 0], B row = [yh | M M M | Cn split in
//     3 | 0] (M = 2^12), Cn = (1 - eps) |c|^2 - A |c|, Rq = tau - (1 - eps) |q|^2 + A |q| + 1, so the accumulator is
//     -2 q.c + Cn - Rq and a pair survives iff its SIGN BIT is set: one v_alignbit per pair collects the bits.
//   * Thresholds in passes over strided subsets of the candidates.  n >= 32768: tau0 from a strided sample of ~4096 rows (exact
//     scan), then the filter over the rows r = 0 mod 16 (which contain the sample), an exact re-rank of those survivors ->
//     tau1 = the exact k-th distance among n / 16 candidates, then the filter over the other 15/16 with tau1: ~16 k survivors
//     per query instead of k n / S (960 at n = 1M), and a 4x smaller sample scan.  n >= 262144: three passes — r = 0 mod 64
//     (sample ~2048 rows), the other multiples of 8, the rest: ~281 survivors per query at 1M.  B rows are stored grouped by
//     residue class so that each pass reads contiguous tiles; each re-rank merges its pass's survivors into the k keys the
//     passes before left (equal keys collapse, so a query that had to re-scan everything in an early pass stays exact).
//   * Data whose scale cannot be normalised (max |x - mu| outside [2^-60, 2^60], or all points identical) pass everything:
//     every query overflows its list and re-scans all candidates — exact, at the scan's speed.
#include <type_traits>

#include "gemm_bf16_tile.h"
#include "knn_fold.h"

namespace {

using namespace dh_bf16;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float widen(unsigned int h) { return __uint_as_float(h << 16); }

// column sums of X in two deterministic stages: partial[b][t] over the rows b, b + gridDim.x, ... ; mu[t] = sum_b / n
__global__ __launch_bounds__(256) void knn_colsum_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                         float* __restrict__ partial) {
  for (int64_t t = threadIdx.x; t < d; t += 256) {
    float s = 0.f;
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) s += X[r * ldx + t];
    partial[(int64_t)blockIdx.x * d + t] = s;
  }
}
__global__ __launch_bounds__(256) void knn_mean_kernel(int64_t n, int64_t d, int n_partial, const float* __restrict__ partial,
                                                       float* __restrict__ mu) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= d) return;
  float s = 0.f;
  for (int b = 0; b < n_partial; ++b) s += partial[(int64_t)b * d + t];
  mu[t] = s / (float)n;
}

// x' = x - mu; norms[r] = sum_t x'[r][t]^2 (one wavefront per row); A2[r] = [hi | hi | lo | 0], B2[r] = [hi | lo | hi | 0] of
// x', each part dp wide, rows K3 = roundup(3 dp, 16) long.
__global__ __launch_bounds__(256) void knn_split_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                        const float* __restrict__ mu, int dp, int64_t K3,
                                                        uint16_t* __restrict__ A2, uint16_t* __restrict__ B2,
                                                        float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float* x = X + r * ldx;
  uint16_t* a = A2 + r * K3;
  uint16_t* b = B2 + r * K3;
  for (int t = 3 * dp + lane; t < K3; t += 64) a[t] = b[t] = 0;
  double s = 0.0;  // the norms are exact up to their final rounding: no d u term in the budget
  for (int t = lane; t < dp; t += 64) {
    const float v = t < d ? x[t] - mu[t] : 0.f;
    s += (double)v * (double)v;
    const unsigned int hi = f32_to_bf16(v);
    const unsigned int lo = f32_to_bf16(v - widen(hi));  // exact subtraction: hi is v rounded to 8 bits
    a[t] = (uint16_t)hi; a[dp + t] = (uint16_t)hi; a[2 * dp + t] = (uint16_t)lo;
    b[t] = (uint16_t)hi; b[dp + t] = (uint16_t)lo; b[2 * dp + t] = (uint16_t)hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) norms[r] = (float)s;
}

// The filter test  |q|^2 + |c|^2 - 2 dot <= tau (1 + chain_rel) + eps (|q|^2 + |c|^2)  rearranged so that the epilogue is one fma
// and one compare per pair:   fma(-2, dot, Cn[c]) <= Rq[q],   Cn = (1 - eps) |c|^2,   Rq = tau (1 + chain_rel) - (1 - eps) |q|^2.
__global__ __launch_bounds__(256) void knn_thresholds_kernel(int64_t n, int64_t q_begin, int64_t nq, int k, float eps, float chain_rel,
                                                             const float* __restrict__ norms, const float* __restrict__ sample_d2,
                                                             float* __restrict__ Rq, float* __restrict__ Cn, int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) Cn[i] = (1.f - eps) * norms[i];
  if (i < nq) {
    const float tau = sample_d2[i * k + (k - 1)];  // +inf when the sample held < k points
    Rq[i] = fmaf(tau, chain_rel, tau) - (1.f - eps) * norms[q_begin + i];
    counts[i] = 0;
  }
}

// grid.x = query tiles x candidate tiles (candidate tile fastest: consecutive blocks share the query tile)
__global__ __launch_bounds__(256) void knn_filter_kernel(int64_t nq, int64_t n, int64_t K3, const uint16_t* __restrict__ A2,
                                                         const uint16_t* __restrict__ B2, const float* __restrict__ Rq,
                                                         const float* __restrict__ Cn, int32_t* __restrict__ counts,
                                                         int32_t* __restrict__ surv, int cap) {
  extern __shared__ __attribute__((aligned(16))) uint16_t tile_lds[];  // TILE_LDS_ELEMS bf16 (two pipeline stages of nt_tile)
  __shared__ float rq[BM];
  const int64_t tiles_n = (n + BN - 1) / BN;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM, n0 = (int64_t)(blockIdx.x % tiles_n) * BN;
  if (threadIdx.x < BM) rq[threadIdx.x] = (m0 + threadIdx.x < nq) ? Rq[m0 + threadIdx.x] : -__int_as_float(0x7f800000);
  // (visible after the barriers inside nt_tile)
  const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
  float cn[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t c = n0 + wn * 64 + j * 32 + (lane & 31);
    cn[j] = c < n ? Cn[c] : 0.f;
  }
  // pass bits first (branch-free), appends afterwards: an in-line "if (pass) atomicAdd" per element serialises one
  // returning global atomic per hit (see the query-stationary kernel below)
  unsigned long long hits = 0ull;
  nt_tile(nq, n, 0, K3, A2, K3, B2, K3, m0, n0, tile_lds, [&](int64_t m, int64_t c, float dot, int e) {
    const float cnj = ((e >> 4) & 1) ? cn[1] : cn[0];
    hits |= (unsigned long long)(fmaf(-2.f, dot, cnj) <= rq[m - m0] ? 1u : 0u) << e;
  });
  const int wm = threadIdx.x >> 7;
  while (hits) {
    const int e = __ffsll((long long)hits) - 1;
    hits &= hits - 1;
    const int i = e >> 5, j = (e >> 4) & 1, r = e & 15;
    const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int64_t c = n0 + wn * 64 + j * 32 + (lane & 31);
    const int pos = atomicAdd(&counts[m], 1);
    if (pos < cap) surv[m * cap + pos] = (int32_t)c;
  }
}

// ---- d <= 64: fp16, "query-stationary", thresholds in two passes (design and error budget: head of this file) ----------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifdef DH_KNN_ABL
#define DH_KNN_ABL_V DH_KNN_ABL
#else
#define DH_KNN_ABL_V 0
#endif

constexpr int64_t FOLD_TWO_PASS_MIN = 32768;   // below this one pass over everything
constexpr int64_t FOLD_THREE_PASS_MIN = 262144;
constexpr int FOLD_NT = 4;                     // 128-row candidate tiles per LDS image (one barrier per image); tile ranges are aligned to it
constexpr int fold_nt(int ks) { return ks >= 5 ? 2 : 4; }  // (K3 = 80: four tiles x two stages would not fit the 160 KB)
constexpr float FOLD_MUL = 4096.f;             // the threshold terms ride as MUL * (t0 + t1 + t2), t_i fp16
constexpr float FOLD_RQ_MAX = 67108864.f;      // 2^26 > any -2 q.c + Cn of scaled rows (<= 3 d 2^18): "everything passes"

// power-of-two scale that puts max |x - mu| into [2^8, 2^9); ok = false when it cannot (then every pair passes)
struct FoldScale { float s; bool ok; };
__device__ __forceinline__ FoldScale fold_scale(unsigned int maxbits) {
  const int e = (int)(maxbits >> 23);  // bits of a non-negative float
  FoldScale r;
  r.ok = e >= 127 - 60 && e <= 127 + 60;
  r.s = r.ok ? __uint_as_float((unsigned int)(127 + 8 - (e - 127)) << 23) : 1.f;
  return r;
}

// Rows are grouped by residue class c = r % G so that every pass reads contiguous tiles of B2: slot 0 = class 0 (pass 1), slots
// 1 .. H - 1 = the other multiples of G / H (the middle pass, H > 1 only), then the rest; row p of B2 = slot p / n1, index p % n1,
// candidate id = index * G + class(slot).  qmagic = ceil(2^16 / (G / H - 1)) divides the small slot numbers exactly (host-checked).
__device__ __forceinline__ unsigned int fold_class_of_slot(unsigned int slot, unsigned int H, unsigned int GH, unsigned int qmagic) {
  const unsigned int m = slot - H;
  return slot < H ? slot * GH : m + ((m * qmagic) >> 16) + 1u;
}

// w ~ t0 + t1 + t2 in fp16 (|w| <= 2^14); residual <= 2^-33 |w|, or 2^-14 where a term falls below the normal range
__device__ __forceinline__ void split3h(float w, _Float16 out[3]) {
  out[0] = (_Float16)w;
  const float r1 = w - (float)out[0];
  out[1] = (_Float16)r1;
  out[2] = (_Float16)(r1 - (float)out[1]);
}

__global__ __launch_bounds__(256) void knn_maxabs_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ mu, unsigned int* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  float m = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4)
    for (int64_t t = lane; t < d; t += 64) m = fmaxf(m, fabsf(X[r * ldx + t] - mu[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order like their bits
}

// One wavefront per row p < n_pad of B2 (slot p / n1, index p % n1; rows without a candidate are zero and never pass).
// y = s (x[r] - mu), norms[r] = |y|^2, A2[r] = [-2 yh | 0 0 0 (-Rq: thresholds kernel) | M M M | 0], B2[p] = [yh | M M M | Cn split | 0].
__global__ __launch_bounds__(256) void knn_fold_split_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ mu, const unsigned int* __restrict__ maxabs,
                                                             int dp, int K3, float eps, float abs_lin, int G, int H, unsigned int qmagic,
                                                             int64_t n1, int64_t n_pad, _Float16* __restrict__ A2,
                                                             _Float16* __restrict__ B2, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pad) return;
  _Float16* b = B2 + p * K3;
  const int64_t r = p < (int64_t)G * n1 ? (p % n1) * G + fold_class_of_slot((unsigned int)(p / n1), H, G / H, qmagic) : n;
  if (r >= n) {
    for (int t = lane; t < K3; t += 64) b[t] = (_Float16)0.f;
    return;
  }
  const FoldScale sc = fold_scale(*maxabs);
  const float* x = X + r * ldx;
  _Float16* a = A2 + r * K3;
  for (int t = dp + 6 + lane; t < K3; t += 64) a[t] = b[t] = (_Float16)0.f;
  float s = 0.f;
  for (int t = lane; t < dp; t += 64) {
    const float y = (t < d && sc.ok) ? (x[t] - mu[t]) * sc.s : 0.f;
    s = fmaf(y, y, s);
    const _Float16 h = (_Float16)y;
    b[t] = h;
    a[t] = (_Float16)(-2.f * (float)h);  // exact
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    norms[r] = s;
    _Float16 c3[3];
    split3h(((1.f - eps) * s - abs_lin * sqrtf(s)) * (1.f / FOLD_MUL), c3);
    for (int i = 0; i < 3; ++i) {
      a[dp + i] = (_Float16)0.f;
      a[dp + 3 + i] = (_Float16)FOLD_MUL;
      b[dp + i] = (_Float16)FOLD_MUL;
      b[dp + 3 + i] = c3[i];
    }
  }
}

// -Rq of every query into the three threshold columns of its A row; tau[i * k + k - 1] = the raw k-th distance so far
__global__ __launch_bounds__(256) void knn_fold_thresholds_kernel(int64_t q_begin, int64_t nq, int k, float eps, float abs_lin,
                                                                  const unsigned int* __restrict__ maxabs,
                                                                  const float* __restrict__ norms, const float* __restrict__ tau,
                                                                  int dp, int K3, _Float16* __restrict__ A2,
                                                                  int32_t* __restrict__ counts, int n_zero_seg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  const FoldScale sc = fold_scale(*maxabs);
  const float tau_s = (tau[i * k + (k - 1)] * sc.s) * sc.s;  // +inf when fewer than k points were seen
  const float nq2 = norms[q_begin + i];
  float rq = tau_s * (1.f + 9.5367431640625e-7f) - (1.f - eps) * nq2 + abs_lin * sqrtf(nq2) + 1.f;
  if (!sc.ok || !(rq < FOLD_RQ_MAX)) rq = FOLD_RQ_MAX;
  _Float16 r3[3];
  split3h(-rq * (1.f / FOLD_MUL), r3);
  _Float16* a = A2 + (q_begin + i) * K3 + dp;
  for (int j = 0; j < 3; ++j) a[j] = r3[j];
  for (int sg = 0; sg < n_zero_seg; ++sg) counts[(int64_t)sg * nq + i] = 0;
}

// A block of 8 waves owns 256 queries for its whole life: each wave keeps the MFMA A-fragments of its 64 queries (all of
// K3) in REGISTERS and the block streams images of FOLD_NT 128-candidate tiles of B2 through a double-buffered LDS copy (one
// barrier per image) — 0.5 bytes of L2 traffic per pair at K3 = 64.  The pass test is ROTATED half a tile against the
// MFMAs: while the matrix cores work on candidate sub-tile j the vector ALUs collect the sign bits of sub-tile 1 - j of
// the previous half, a fixed number of pairs behind every MFMA (order written out and pinned: left to itself the compiler
// sinks the bit collection into the rarely taken append branch behind the MFMAs).
// History (profiles/): three-term bf16 form of this kernel, K3 = 176, v_cmp + v_addc per pair: 353 ms at 1M x 50, of which the
// survivor appends (~960 per query) cost 81 ms; the tile loop sat at ~59 % matrix-pipe utilisation.
template <int KS, int NT>  // 16-wide k steps: K3 = 16 KS; NT tiles per LDS image
__global__ __launch_bounds__(512) void knn_fold_filter_kernel(int64_t nq, int64_t n, int G, int H, unsigned int qmagic, int64_t n1,
                                                              const _Float16* __restrict__ A2, const _Float16* __restrict__ B2,
                                                              int32_t* __restrict__ counts, int32_t* __restrict__ surv, int cap, int seg,
                                                              int64_t t_begin, int64_t t_end, int64_t tiles_per_slice) {
  constexpr int K3 = 16 * KS;
  constexpr int LD = K3 + 8;               // fp16 per LDS row: (LD / 2) % 8 == 4 -> conflict-free 16-byte fragment reads
  constexpr int CPR = 2 * KS;              // 16-byte chunks per row
  constexpr int GCH = NT * BN * CPR;       // chunks per image (contiguous in B2: a row is exactly CPR chunks)
  constexpr int NCH = (GCH + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) _Float16 lds_h[];  // 2 x [NT * BN][LD]
  __shared__ int cnt_s[256];  // survivors appended by THIS block per query: the block owns its 256 queries, so the
                              // list cursors are LDS atomics instead of returning global atomics
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, kh = (lane >> 5) * 8;
  const int64_t m0 = (int64_t)blockIdx.x * 256 + wr * 64;

  if (tid < 256) cnt_s[tid] = 0;
  f16x8 a[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t row = m0 + i * 32 + lr;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      u32x4 v = u32x4(0u);
      if (row < nq) v = *reinterpret_cast<const u32x4*>(A2 + row * K3 + kk * 16 + kh);
      a[i][kk] = __builtin_bit_cast(f16x8, v);
    }
  }

  int* cnt = cnt_s + wr * 64 + 4 * (lane >> 5);
  const int64_t row0 = m0 + 4 * (lane >> 5);                           // query of accumulator register (i, r):
  int32_t* surv_base = surv + row0 * cap + (int64_t)blockIdx.y * seg;  // row0 + i * 32 + (r & 3) + 8 * (r >> 2); this slice's segment

  const int64_t t_lo = t_begin + (int64_t)blockIdx.y * tiles_per_slice, t_hi = min(t_end, t_lo + tiles_per_slice);
  u32x4 stage[NCH];
  const _Float16* b_src = B2 + (int64_t)tid * 8;  // chunk `tid` of image 0
  int lds_off[NCH];
#pragma unroll
  for (int s = 0; s < NCH; ++s) {
    const int c = tid + 512 * s;
    lds_off[s] = (c / CPR) * LD + (c % CPR) * 8;
  }
  // B2 is padded with zero rows to whole images: no guards
  static_assert(GCH % 512 == 0, "an image is a whole number of 512-thread rounds");
  auto load_image = [&](int64_t t) __attribute__((always_inline)) {
    const _Float16* src = b_src + t * BN * (int64_t)K3;
#pragma unroll
    for (int s = 0; s < NCH; ++s) stage[s] = *reinterpret_cast<const u32x4*>(src + (int64_t)512 * 8 * s);
  };
  auto store_image = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NCH; ++s) *reinterpret_cast<u32x4*>(lds_h + (size_t)buf * NT * BN * LD + lds_off[s]) = stage[s];
  };
  f32x16 acc[2][2];
  const unsigned int n1u = (unsigned int)n1;
  const int rows_left = (int)min((int64_t)64, nq - row0);  // queries of this lane's accumulator rows that exist
  // Only lanes with a hit walk their set bits.
  // (g0, j0): slot and in-class index of the first row of the candidate tile; a tile crosses at most one slot boundary
  // (n1 >= 128 whenever G > 1)
  auto append = [&](int j, unsigned int g0, unsigned int j0, const unsigned int (&h)[2]) __attribute__((always_inline)) {
#if defined(DH_KNN_ABL)  // ablation builds (timing only, results are garbage): 1 no appends, 2 + no barrier, 3 + no image traffic
    if (h[0] == 0xdeadbeefu && h[1] == 0x12345u) cnt[0] = 1;  // ablation build: keep h alive, never append
    return;
#endif
    if (__builtin_amdgcn_ballot_w64((h[0] | h[1]) != 0u) == 0ull) return;  // nothing in this wave: skip the id arithmetic too
    unsigned int jj = j0 + (unsigned int)(wc * 64 + j * 32 + lr), gg = g0;
    if (jj >= n1u) { jj -= n1u; gg += 1u; }
    const unsigned int id = jj * (unsigned int)G + fold_class_of_slot(gg, (unsigned int)H, (unsigned int)(G / H), qmagic);
    const bool live = gg < (unsigned int)G && (int64_t)id < n;  // the zero rows behind the last candidate never pass anyway
    // one mask for both query sub-tiles: bit 16 i + (15 - r) = accumulator register r of sub-tile i; one hit per round (lanes
    // rarely hold two), one returning LDS atomic + one 4-byte store each
    unsigned int mk = live ? (h[0] | (h[1] << 16)) : 0u;
    while (mk) {
      const int b = 31 - __clz(mk);
      mk ^= 1u << b;
      const int r = 15 - (b & 15);
      const int q = (b >> 4) * 32 + (r & 3) + 8 * (r >> 2);
      if (q < rows_left) {  // rows beyond the query range hold zero fragments (accumulator +0: no pass), checked all the same
        const int pos = atomicAdd(cnt + q, 1);
        if (pos < seg) surv_base[(unsigned int)(q * cap + pos)] = (int32_t)id;
      }
    }
  };
  // The B fragments of a half tile are read from LDS during the MFMAs of the half before it (two register sets); only the
  // first half of an image waits for its reads.
  auto load_frags = [&](f16x8 (&dst)[KS], const _Float16* p) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) dst[kk] = *reinterpret_cast<const f16x8*>(p + kk * 16);
  };
  auto half_tile = [&](auto jm_tag, const f16x8 (&b)[KS], f16x8 (&b_next)[KS], const _Float16* next_frag, unsigned int (&h)[2],
                       int64_t prefetch_tile) __attribute__((always_inline)) {
    constexpr int JM = decltype(jm_tag)::value, JT = 1 - JM;
    constexpr int PER = (32 + 2 * KS - 1) / (2 * KS);  // pairs behind each MFMA
    h[0] = h[1] = 0u;
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
      const int kk = m >> 1, i = m & 1;
      if (kk == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        acc[i][JM] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][kk], b[kk], z, 0, 0, 0);
      } else {
        acc[i][JM] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][kk], b[kk], acc[i][JM], 0, 0, 0);
      }
      if (m == 0 && next_frag) load_frags(b_next, next_frag);
#pragma unroll
      for (int e = m * PER; e < (m + 1) * PER && e < 32; ++e)
        h[e >> 4] = __builtin_amdgcn_alignbit(h[e >> 4], __float_as_uint(acc[e >> 4][JT][e & 15]), 31u);
      __builtin_amdgcn_sched_barrier(0);
      // the next image's loads go out behind the FIRST MFMA of the image: the compiler drains vmcnt (the previous half's
      // append stores share the counter with loads) in front of this block, so this is the earliest point that keeps them in flight
      if (JM == 0 && m == 0 && prefetch_tile >= 0 && !(DH_KNN_ABL_V >= 3)) {
        load_image(prefetch_tile);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  load_image(t_lo < t_hi ? t_lo : 0);
  store_image(0);
  __syncthreads();
  int cur = 0;
  unsigned int h[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][1][r] = 1.f;  // nothing passes in the first rotated half
  unsigned int g_cur = (unsigned int)((t_lo * BN) / n1), j_cur = (unsigned int)((t_lo * BN) % n1), g_prev = 0u, j_prev = 0u;
  f16x8 bq[2][KS];
  int64_t t = t_lo;
  for (; t < t_hi; t += NT) {
    const _Float16* img = lds_h + (size_t)cur * NT * BN * LD + (wc * 64 + lr) * LD + kh;  // + (sub * BN + JM * 32) * LD
    load_frags(bq[0], img);
#pragma unroll
    for (int sub = 0; sub < NT; ++sub) {
      // half A: matrix cores on sub-tile 0 of tile t + sub, vector ALUs on sub-tile 1 of the tile before it
      __builtin_amdgcn_sched_barrier(0);
      half_tile(std::integral_constant<int, 0>{}, bq[0], bq[1], img + (sub * BN + 32) * LD, h, (sub == 0 && t + NT < t_hi) ? t + NT : -1);
      append(1, g_prev, j_prev, h);
      // half B: matrix cores on sub-tile 1, vector ALUs on sub-tile 0 of the same tile
      __builtin_amdgcn_sched_barrier(0);
      half_tile(std::integral_constant<int, 1>{}, bq[1], bq[0], sub + 1 < NT ? img + (sub + 1) * BN * LD : nullptr, h, -1);
      append(0, g_cur, j_cur, h);
      g_prev = g_cur; j_prev = j_cur;
      j_cur += BN;
      if (j_cur >= n1u) { j_cur -= n1u; g_cur += 1u; }
    }
#if defined(DH_KNN_ABL) && DH_KNN_ABL >= 3
    if (t_hi == -12345) store_image(cur ^ 1);
#else
    if (t + NT < t_hi) store_image(cur ^ 1);
#endif
#if defined(DH_KNN_ABL) && DH_KNN_ABL >= 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    __syncthreads();  // everyone is done with `cur` and the next image is complete
#endif
    cur ^= 1;
  }
  if (t_hi > t_lo) {  // the last half's pass bits
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      h[i] = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) h[i] = h[i] + h[i] + (__float_as_uint(acc[i][1][r]) >> 31);
    }
    append(1, g_prev, j_prev, h);
  }
  __syncthreads();
  if (tid < 256 && (int64_t)blockIdx.x * 256 + tid < nq) counts[(int64_t)blockIdx.y * nq + (int64_t)blockIdx.x * 256 + tid] = cnt_s[tid];
}

// One wavefront per query: exact chain distances of its survivors (or of every candidate after an overflow), k smallest
// (d2, index) keys by k rounds of wave-wide minimum.  Keys are (bits(d2) << 32 | index): d2 >= +0, so the unsigned order
// of the bits is the order of the values and ties fall to the lower index.
constexpr int RR_CHUNK = 1024;  // keys held in LDS per wavefront and round

template <bool VEC>
__device__ __forceinline__ float chain_d2(const float* __restrict__ xq, const float* __restrict__ xc, int64_t d) {
  float acc = 0.f;
  if constexpr (VEC) {  // 16-byte aligned rows, d % 4 == 0
    // The chain is sequential by definition, its LOADS are not: written as one loop over 16-byte pieces this was a round trip per four
    // features — 13 in a row for a 50-d survivor (ISA: two loads, s_waitcnt vmcnt(0), eight flops, branch), which is what the re-rank
    // passes' 8 ms each at 1M x 50 were.  Pieces are requested 8, then 4, then 1 at a time and added in index order (same bits).
    int64_t t = 0;
    auto piece = [&](auto n_c) __attribute__((always_inline)) {
      constexpr int NP = decltype(n_c)::value;
      for (; t + 4 * NP <= d; t += 4 * NP) {
        f32x4 a[NP], b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          a[p] = *reinterpret_cast<const f32x4*>(xq + t + 4 * p);
          b[p] = *reinterpret_cast<const f32x4*>(xc + t + 4 * p);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float diff = __fsub_rn(a[p][u], b[p][u]);
            acc = __fadd_rn(acc, __fmul_rn(diff, diff));
          }
      }
    };
    piece(std::integral_constant<int, 8>{});
    piece(std::integral_constant<int, 4>{});
    piece(std::integral_constant<int, 1>{});
  } else {
    for (int64_t t = 0; t < d; ++t) {
      const float diff = __fsub_rn(xq[t], xc[t]);
      acc = __fadd_rn(acc, __fmul_rn(diff, diff));
    }
  }
  return acc;
}

// Survivor list of query q in this pass: n_seg segments of seg slots at surv + q * cap (the caller has added the pass's offset);
// counts[s * nq + q] entries of segment s are valid.  Any segment over capacity: the
// list is incomplete -> every candidate is re-scanned.  carry_in: the k keys (out_idx, raw d2 in out_dist) an earlier pass
// left take part (equal keys collapse in the selection, so candidates seen twice are harmless).  raw_out: leave raw d2.
template <bool VEC>
__global__ __launch_bounds__(256) void knn_rerank_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                         int64_t q_begin, int64_t nq, int k, const int32_t* __restrict__ counts,
                                                         const int32_t* __restrict__ surv, int cap, int n_seg, int seg,
                                                         int carry_in, int raw_out,
                                                         int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  if (q >= nq) return;
  // per wavefront: keys[0 .. RR_CHUNK) = this round's candidates, keys[RR_CHUNK .. RR_CHUNK + k) = best so far
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem) + (size_t)wave * (RR_CHUNK + 64);
  const unsigned long long kInf = ~0ull;
  const float* xq = X + (q_begin + q) * ldx;
  bool overflow = false;
  for (int sg = 0; sg < n_seg; ++sg) overflow |= counts[(int64_t)sg * nq + q] > seg;
  int carried = 0;
  if (carry_in) {
    const bool have = lane < k && out_idx[q * k + lane] >= 0;  // a valid prefix (ascending keys, then -1 entries)
    if (have) keys[RR_CHUNK + lane] = ((unsigned long long)__float_as_uint(out_dist[q * k + lane]) << 32) | (unsigned int)out_idx[q * k + lane];
    carried = __popcll(__ballot(have));
    __builtin_amdgcn_wave_barrier();
  }
  for (int sg = 0; sg < (overflow ? 1 : n_seg); ++sg) {
  const int64_t total = overflow ? n : counts[(int64_t)sg * nq + q];
  const int32_t* mine = surv + q * cap + (int64_t)sg * seg;
  for (int64_t base = 0; base < total; base += RR_CHUNK) {
    const int m = (int)min((int64_t)RR_CHUNK, total - base);
    for (int i = lane; i < m; i += 64) {
      const int c = overflow ? (int)(base + i) : mine[base + i];
      const float d2 = chain_d2<VEC>(xq, X + (int64_t)c * ldx, d);
      keys[i] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)c;
    }
    const unsigned long long carry = lane < carried ? keys[RR_CHUNK + lane] : kInf;  // snapshot (k <= 64)
    __builtin_amdgcn_wave_barrier();
    // k rounds of "smallest key above the previous pick" over this round's keys and the carried ones
    unsigned long long last = 0;
    int found = 0;
    for (int s = 0; s < k; ++s) {
      unsigned long long best = (s == 0 || carry > last) ? carry : kInf;
      for (int i = lane; i < m; i += 64) {
        const unsigned long long v = keys[i];
        if ((s == 0 || v > last) && v < best) best = v;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other < best ? other : best;
      }
      if (best == kInf) break;
      if (lane == 0) keys[RR_CHUNK + s] = best;
      last = best;
      ++found;
    }
    carried = found;
    __builtin_amdgcn_wave_barrier();
  }
  }
  for (int s = lane; s < k; s += 64) {
    if (s < carried) {
      const unsigned long long v = keys[RR_CHUNK + s];
      out_idx[q * k + s] = (int32_t)(unsigned int)v;
      const float d2 = __uint_as_float((unsigned int)(v >> 32));
      out_dist[q * k + s] = raw_out ? d2 : (float)sqrt((double)d2);
    } else {
      out_idx[q * k + s] = -1;
      out_dist[q * k + s] = __int_as_float(0x7f800000);
    }
  }
}

// sample rows j * stride of X, zero-padded to `rs` columns
__global__ __launch_bounds__(256) void knn_sample_kernel(int64_t S, int64_t stride, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                         int rs, float* __restrict__ Xs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S * rs) return;
  const int64_t j = i / rs;
  const int t = (int)(i % rs);
  Xs[i] = t < d ? X[j * stride * ldx + t] : 0.f;
}

}  // namespace

namespace dh {

constexpr int64_t kMeanBlocks = 1024;
size_t knn_filter_mean_floats(int64_t d) { return (size_t)(kMeanBlocks + 1) * (size_t)d; }  // mu[d] + partial[1024][d]

int64_t knn_filter_sample_size(int64_t n) {
  int64_t S = n / 64;
  if (S < 4096) S = 4096;
  if (S > 32768) S = 32768;
  return S < n ? S : n;
}

static int pow2_at_least(int64_t want, int lo, int hi) {
  int cap = lo;
  while (cap < want && cap < hi) cap <<= 1;
  return cap;
}

int knn_filter_cap(int64_t n, int k) {
  // expected survivors ~ k n / S (times the eps margin); 4x that, at least 256, a power of two
  return pow2_at_least(4 * (int64_t)k * dh::ceil_div(n, knn_filter_sample_size(n)), 256, 1 << 16);
}

// bf16 columns of an operand row of the tile kernel (d > 64): three dp-wide parts, whole 16-wide MFMA steps
int64_t knn_filter_k3(int64_t d) { return (3 * (int64_t)((d + 7) / 8 * 8) + 15) / 16 * 16; }

// Survivor lists of the tile kernel (d > 64): one segment per query (an overfull list only costs that query a re-scan).
void knn_filter_geometry(int64_t n, int64_t d, int64_t nq, int k, int* n_seg, int* seg) {
  (void)d; (void)nq;
  *n_seg = 1;
  *seg = knn_filter_cap(n, k);
}

int knn_filter_padded_d(int64_t d) { return (int)((d + 7) / 8 * 8); }

void knn_filter_sample_strided(int64_t S, int64_t stride, int64_t d, const float* X, int64_t ldx, int rs, float* Xs, hipStream_t st) {
  hipLaunchKernelGGL(knn_sample_kernel, dim3((unsigned)ceil_div(S * rs, 256)), dim3(256), 0, st, S, stride, d, X, ldx, rs, Xs);
}
void knn_filter_sample(int64_t n, int64_t d, const float* X, int64_t ldx, int rs, float* Xs, hipStream_t st) {
  const int64_t S = knn_filter_sample_size(n);
  knn_filter_sample_strided(S, n / S, d, X, ldx, rs, Xs, st);
}

static void colmeans(int64_t n, int64_t d, const float* X, int64_t ldx, float* mean_ws, hipStream_t st) {
  const int n_partial = (int)(n < kMeanBlocks ? n : kMeanBlocks);
  hipLaunchKernelGGL(knn_colsum_kernel, dim3((unsigned)n_partial), dim3(256), 0, st, n, d, X, ldx, mean_ws + d);
  hipLaunchKernelGGL(knn_mean_kernel, dim3((unsigned)ceil_div(d, 256)), dim3(256), 0, st, n, d, n_partial, mean_ws + d, mean_ws);
}

static void rerank_launch(int64_t n, const float* Xr, int64_t ldr, int64_t dr, int64_t q_begin, int64_t nq, int k, const int32_t* counts,
                          const int32_t* surv, int cap, int n_seg, int seg, int carry_in, int raw_out,
                          int32_t* out_idx, float* out_dist, hipStream_t st) {
  const size_t lds = 4 * (size_t)(RR_CHUNK + 64) * sizeof(unsigned long long);
  const bool vec = dr % 4 == 0 && ldr % 4 == 0 && aligned16(Xr);
  if (vec)
    hipLaunchKernelGGL(knn_rerank_kernel<true>, dim3((unsigned)ceil_div(nq, 4)), dim3(256), lds, st, n, dr, Xr, ldr, q_begin, nq, k, counts,
                       surv, cap, n_seg, seg, carry_in, raw_out, out_idx, out_dist);
  else
    hipLaunchKernelGGL(knn_rerank_kernel<false>, dim3((unsigned)ceil_div(nq, 4)), dim3(256), lds, st, n, dr, Xr, ldr, q_begin, nq, k, counts,
                       surv, cap, n_seg, seg, carry_in, raw_out, out_idx, out_dist);
}

// Steps 2 and 3 for d > 64 (the sample pass has already left the raw k-th sample distances in sample_d2 [nq][k]).
// `Xr` (leading dimension ldr, dr >= d columns, extra columns zero) is what the re-rank reads.
int knn_filter_launch(int64_t n, int64_t d, const float* X, int64_t ldx, const float* Xr, int64_t ldr, int64_t dr,
                      int64_t q_begin, int64_t nq, int k,
                      const float* sample_d2, float* mean_ws, uint16_t* A2, uint16_t* B2, float* norms, float* Rq, float* Cn,
                      int32_t* counts, int32_t* surv, int32_t* out_idx, float* out_dist, hipStream_t st) {
  const int dp = knn_filter_padded_d(d);
  const int64_t K3 = knn_filter_k3(d);
  const float eps = 6.103515625e-5f + (float)d * 2.384185791015625e-7f;  // 2^-14 + d 2^-22 (budget at the head of the file)
  const float chain_rel = (float)(d + 16) * 6.3e-8f;                     // (d + 16) u, 5 % over
  colmeans(n, d, X, ldx, mean_ws, st);
  hipLaunchKernelGGL(knn_split_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, n, d, X, ldx, mean_ws, dp, K3, A2, B2, norms);
  hipLaunchKernelGGL(knn_thresholds_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, q_begin, nq, k, eps, chain_rel, norms,
                     sample_d2, Rq, Cn, counts);
  const int cap = knn_filter_cap(n, k);
  const int64_t tiles = ceil_div(nq, BM) * ceil_div(n, BN);
  if (tiles >= (int64_t)1 << 31) return fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: too many filter tiles (%lld)", (long long)tiles);
  constexpr size_t kTileLds = (size_t)TILE_LDS_ELEMS * sizeof(uint16_t);
  static const bool tile_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)kTileLds) == hipSuccess;
  if (!tile_ok) return fail(DH_ERR_LAUNCH, "dh_knn_bruteforce_f32: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(knn_filter_kernel, dim3((unsigned)tiles), dim3(256), kTileLds, st, nq, n, K3, A2 + q_begin * K3, B2, Rq, Cn, counts,
                     surv, cap);
  rerank_launch(n, Xr, ldr, dr, q_begin, nq, k, counts, surv, cap, 1, cap, 0, 0, out_idx, out_dist, st);
  return check_launch("dh_knn_bruteforce_f32(filter)");
}

// ---- d <= 64 ---------------------------------------------------------------------------------------------------------------
bool knn_fold_applies(int64_t d) { return d <= 64; }

KnnFoldGeom knn_fold_geom(int64_t n, int64_t d, int64_t nq, int k) {
  KnnFoldGeom g{};
  g.dp = knn_filter_padded_d(d);
  g.K3 = (g.dp + 6 + 15) / 16 * 16;
  // thresholds in up to three passes: every 64th row, then the other multiples of 8, then the rest (n >= 2^18); every 16th
  // row, then the rest (n >= 2^15); else one pass
  int64_t sample_target;
  if (n >= FOLD_THREE_PASS_MIN) { g.G = 64; g.H = 8; g.n_pass = 3; sample_target = 2048; }
  else if (n >= FOLD_TWO_PASS_MIN) { g.G = 16; g.H = 1; g.n_pass = 2; sample_target = 4096; }
  else { g.G = 1; g.H = 1; g.n_pass = 1; sample_target = 0; }
  {  // exact division of the slot numbers by G / H - 1 through a 16-bit reciprocal (checked here once)
    const unsigned int q = (unsigned int)(g.G / g.H) > 1u ? (unsigned int)(g.G / g.H) - 1u : 1u;
    g.qmagic = (65536u + q - 1u) / q;
    for (unsigned int m = 0; m < (unsigned int)g.G; ++m)
      if (((m * g.qmagic) >> 16) != m / q) { g.qmagic = 0; break; }  // (never: G <= 64)
  }
  g.n1 = ceil_div(n, g.G);
  g.n_pos = ceil_div(g.G * g.n1, (int64_t)FOLD_NT * BN) * FOLD_NT * BN;  // zero rows up to whole LDS images
  if (g.G > 1) {  // every sample row lies in pass 1 (stride a multiple of G)
    int64_t m = (n + g.G * sample_target / 2) / (g.G * sample_target);
    if (m < 1) m = 1;
    g.stride0 = g.G * m;
    g.S = ceil_div(n, g.stride0);
  } else {
    g.S = knn_filter_sample_size(n);
    g.stride0 = n / g.S;
  }
  g.tiles = ceil_div(g.n_pos, BN);
  auto tiles_of = [&](int64_t rows) {
    int64_t t = ceil_div(ceil_div(rows, BN), FOLD_NT) * FOLD_NT;
    return t > g.tiles ? g.tiles : t;
  };
  int64_t bound[4] = {0, g.tiles, g.tiles, g.tiles};
  if (g.n_pass >= 2) bound[1] = tiles_of(g.n1);
  if (g.n_pass == 3) bound[2] = tiles_of(g.H * g.n1);
  // candidate slices (grid.y) until >= 2 rounds of blocks exist; each slice owns a segment of every survivor list, at least
  // 1024 slots deep so that a query whose neighbours all sit in one slice does not overflow
  const int64_t qblocks = ceil_div(nq, 256);
  const int64_t slices = qblocks >= 512 ? 1 : ceil_div(512, qblocks);
  g.cap = 0;
  for (int ps = 0; ps < g.n_pass; ++ps) {
    g.t_begin[ps] = bound[ps];
    g.t_end[ps] = bound[ps + 1];
    const int64_t T = g.t_end[ps] - g.t_begin[ps];
    // expected survivors: pass 1 ~ k (its rows) / S, 4x that; later passes: the threshold is the exact k-th distance among the
    // fraction f of the rows seen so far -> ~ k / f candidates in all, 8x that (the fp16 margin inflates these lists first)
    int64_t want;
    if (ps == 0) want = 4 * (int64_t)k * ceil_div(T * BN, g.S);
    else if (ps == 1 && g.n_pass == 3) want = 8 * (int64_t)k * g.H;
    else want = 8 * (int64_t)k * (g.G / g.H);
    if (T <= 0) { g.tps[ps] = FOLD_NT; g.n_seg[ps] = 0; g.seg[ps] = 0; continue; }
    int64_t sl = slices;
    if (sl > ceil_div(T, FOLD_NT)) sl = ceil_div(T, FOLD_NT);
    g.tps[ps] = (T / sl) / FOLD_NT * FOLD_NT;  // rounded DOWN to whole images: at least as many slices as asked for
    if (g.tps[ps] < FOLD_NT) g.tps[ps] = FOLD_NT;
    g.n_seg[ps] = (int)ceil_div(T, g.tps[ps]);
    const int cap = pow2_at_least(want, 256, 1 << 16);
    int sg = cap / g.n_seg[ps];
    if (g.n_seg[ps] > 1 && sg < 1024) sg = cap < 1024 ? cap : 1024;
    g.seg[ps] = sg;
    g.cap += g.n_seg[ps] * sg;
  }
  g.n_seg_total = g.n_seg[0] + g.n_seg[1] + g.n_seg[2];
  return g;
}

int knn_fold_launch(const KnnFoldGeom& g, int64_t n, int64_t d, const float* X, int64_t ldx, const float* Xr, int64_t ldr, int64_t dr,
                    int64_t q_begin, int64_t nq, int k, float* mean_ws, unsigned int* maxabs, void* A2v, void* B2v, float* norms,
                    int32_t* counts, int32_t* surv, int32_t* out_idx, float* out_dist, hipStream_t st) {
  _Float16* A2 = static_cast<_Float16*>(A2v);
  _Float16* B2 = static_cast<_Float16*>(B2v);
  if (g.qmagic == 0) return fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: slot reciprocal");
  const float eps = 9.765625e-4f + 1.220703125e-4f + (float)g.dp * 9.5367431640625e-7f;  // 2^-10 + 2^-13 + dp 2^-20
  const float abs_lin = 1.01f * 1.220703125e-4f * sqrtf((float)g.dp);                    // 1.01 * 2^-13 sqrt(dp)
  colmeans(n, d, X, ldx, mean_ws, st);
  if (dh::zero_async(maxabs, sizeof(unsigned int), st) != hipSuccess) return fail(DH_ERR_LAUNCH, "dh_knn_bruteforce_f32: memset failed");
  hipLaunchKernelGGL(knn_maxabs_kernel, dim3((unsigned)(n < 4096 ? ceil_div(n, 4) : 1024)), dim3(256), 0, st, n, d, X, ldx, mean_ws, maxabs);
  hipLaunchKernelGGL(knn_fold_split_kernel, dim3((unsigned)ceil_div(g.n_pos, 4)), dim3(256), 0, st, n, d, X, ldx, mean_ws, maxabs, g.dp, g.K3,
                     eps, abs_lin, g.G, g.H, g.qmagic, g.n1, g.n_pos, A2, B2, norms);
  const dim3 tgrid((unsigned)ceil_div(nq, 256));
  const int64_t qblocks = ceil_div(nq, 256);
  const int ks = g.K3 / 16;
  const size_t lds = 2 * (size_t)fold_nt(ks) * BN * (g.K3 + 8) * sizeof(_Float16);
  auto filter = [&](int n_seg, int seg, int64_t t_begin, int64_t t_end, int64_t tps, int32_t* cnt, int32_t* sv) -> int {
    if (n_seg <= 0) return DH_OK;
    dim3 grid((unsigned)qblocks, (unsigned)n_seg);
#define DH_KNN_FS(KS)                                                                                                          \
  case KS: {                                                                                                                   \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_fold_filter_kernel<KS, fold_nt(KS)>),         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,                                     \
                                               (int)(2 * fold_nt(KS) * BN * (16 * KS + 8) * sizeof(_Float16))) == hipSuccess;  \
    if (!ok) return fail(DH_ERR_LAUNCH, "dh_knn_bruteforce_f32: cannot raise the dynamic LDS limit");                        \
    hipLaunchKernelGGL((knn_fold_filter_kernel<KS, fold_nt(KS)>), grid, dim3(512), lds, st, nq, n, g.G, g.H, g.qmagic, g.n1,   \
                       A2 + q_begin * g.K3, B2, cnt, sv, g.cap, seg, t_begin, t_end, tps);                                     \
  } break
    switch (ks) {
      DH_KNN_FS(1); DH_KNN_FS(2); DH_KNN_FS(3); DH_KNN_FS(4); DH_KNN_FS(5);
      default: return fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: unexpected filter depth %d", ks);
    }
#undef DH_KNN_FS
    return DH_OK;
  };
  // per pass: thresholds from the k-th distance so far (out_dist, raw d2) -> filter over the pass's tiles -> exact k smallest of
  // its survivors merged into the keys the passes before left (out_idx / out_dist; raw d2 until the last pass)
  int64_t seg_base = 0, slot_base = 0;
  for (int ps = 0; ps < g.n_pass; ++ps) {
    if (g.n_seg[ps] <= 0) continue;
    hipLaunchKernelGGL(knn_fold_thresholds_kernel, tgrid, dim3(256), 0, st, q_begin, nq, k, eps, abs_lin, maxabs, norms, out_dist, g.dp, g.K3, A2,
                       counts, ps == 0 ? g.n_seg_total : 0);
    int32_t* cnt = counts + seg_base * nq;
    int32_t* sv = surv + slot_base;
    const int rc = filter(g.n_seg[ps], g.seg[ps], g.t_begin[ps], g.t_end[ps], g.tps[ps], cnt, sv);
    if (rc != DH_OK) return rc;
    bool last = true;
    for (int nx = ps + 1; nx < g.n_pass; ++nx) last = last && g.n_seg[nx] <= 0;
    rerank_launch(n, Xr, ldr, dr, q_begin, nq, k, cnt, sv, g.cap, g.n_seg[ps], g.seg[ps], ps > 0 ? 1 : 0, last ? 0 : 1, out_idx, out_dist, st);
    seg_base += g.n_seg[ps];
    slot_base += (int64_t)g.n_seg[ps] * g.seg[ps];
  }
  return check_launch("dh_knn_bruteforce_f32(filter)");
}

}  // namespace dh
