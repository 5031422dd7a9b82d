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
//               with K = 3 d).  |d2~ - d2| <= eps (|q|^2 + |c|^2) with eps = 2^-13 + d 2^-20 (derivation below), so the
//               pairs with d2~ <= tau_q + eps (|q|^2 + |c|^2) are a superset of the true neighbours; they are appended
//               to a per-query survivor list (~ k n / S entries).
//   3. rerank = the exact chain on the survivors only, k smallest (d2, index) per query — the same keys, hence the same
//               bits, as the full scan.  A query whose list overflowed re-scans all candidates inside the same kernel.
//
// Error budget of step 2 (u = 2^-24; all worst case, no statistics):
//   representation: dropped lo.lo and r terms            <= 3.1 * 2^-16 sum|q_i c_i|
//   MFMA accumulation over 3d bf16 products (exact each) <= 3d u (1 + 2^-7) sum|q_i c_i|
//   fp32 norms, the final subtraction/addition            <= (d + 6) u (|q|^2 + |c|^2)
//   the chain itself vs the real d2                       <= 2 (d + 2) u (|q|^2 + |c|^2)
// with sum|q_i c_i| <= (|q|^2 + |c|^2) / 2 and the factor 2 in front of the dot product:
//   |d2~ - d2_chain| <= (|q|^2 + |c|^2) (3.1 * 2^-16 + (6d + 10) u)  <  (|q|^2 + |c|^2) (2^-13 + d 2^-20) / 2.
// The margin scales with the NORMS, so the filter works on CENTRED copies x' = fl(x - mu), mu = column means: distances
// are translation invariant (any mu is admissible — its rounding only costs efficiency), the subtraction's own rounding
// moves d2 by <= 2 |x-y| u (|x'| + |y'|) <= 4 u (|x'|^2 + |y'|^2), well inside the slack above, and embeddings with a
// large common offset (or non-negative expression rows) no longer inflate the survivor lists.  Inputs must be finite.
#include <type_traits>

#include "gemm_bf16_tile.h"

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

// exact three-way split of an fp32 value into bf16 terms (24 = 8 + 8 + 8 significant bits): v == hi + mid + lo
__device__ __forceinline__ void split3(float v, unsigned int out[3]) {
  out[0] = f32_to_bf16(v);
  if (!(fabsf(v) < __int_as_float(0x7f800000))) {  // +-inf thresholds (rows that pass everything / nothing): no inf - inf
    out[1] = out[2] = 0u;
    return;
  }
  const float r1 = v - widen(out[0]);
  out[1] = f32_to_bf16(r1);
  out[2] = f32_to_bf16(r1 - widen(out[1]));
}

// x' = x - mu; norms[r] = sum_t x'[r][t]^2 (one wavefront per row); A2[r] = [hi | hi | lo | 0], B2[r] = [hi | lo | hi | 0] of
// x', each part dp wide, rows K3 = roundup(3 dp, 16) long.
// FOLD (query-stationary kernel, d <= 64): the whole filter test rides the matrix cores.  A2[r] = [-2 hi | -2 hi | -2 lo |
// -Rq split3 (written by knn_thresholds_kernel) | 1 1 1 | 0], B2[r] = [hi | lo | hi | 1 1 1 | Cn split3 | 0] with
// Cn = (1 - eps) |x'|^2, so that the accumulator IS  -2 q.c + Cn[c] - Rq[q]  and a pair survives iff it is <= 0.
template <bool FOLD>
__global__ __launch_bounds__(256) void knn_split_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                        const float* __restrict__ mu, int dp, int64_t K3, float eps,
                                                        uint16_t* __restrict__ A2, uint16_t* __restrict__ B2,
                                                        float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float* x = X + r * ldx;
  uint16_t* a = A2 + r * K3;
  uint16_t* b = B2 + r * K3;
  for (int t = 3 * dp + (FOLD ? 6 : 0) + lane; t < K3; t += 64) a[t] = b[t] = 0;
  float s = 0.f;
  for (int t = lane; t < dp; t += 64) {
    const float v = t < d ? x[t] - mu[t] : 0.f;
    s = fmaf(v, v, s);
    const unsigned int hi = f32_to_bf16(v);
    const unsigned int lo = f32_to_bf16(v - widen(hi));  // exact subtraction: hi is v rounded to 8 bits
    // -2 x: exponent + 1 and the sign flipped, exact in bf16 (finite inputs; an overflow to inf only widens the test)
    const unsigned int ahi = FOLD ? f32_to_bf16(-2.f * widen(hi)) : hi, alo = FOLD ? f32_to_bf16(-2.f * widen(lo)) : lo;
    a[t] = (uint16_t)ahi; a[dp + t] = (uint16_t)ahi; a[2 * dp + t] = (uint16_t)alo;
    b[t] = (uint16_t)hi; b[dp + t] = (uint16_t)lo; b[2 * dp + t] = (uint16_t)hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    norms[r] = s;
    if (FOLD) {
      unsigned int c3[3];
      split3((1.f - eps) * s, c3);
      for (int i = 0; i < 3; ++i) {
        a[3 * dp + 3 + i] = 0x3f80;  // 1.0 against Cn
        b[3 * dp + i] = 0x3f80;      // 1.0 against -Rq
        b[3 * dp + 3 + i] = (uint16_t)c3[i];
        a[3 * dp + i] = 0;           // -Rq: knn_thresholds_kernel (query rows only)
      }
    }
  }
}

// The filter test  |q|^2 + |c|^2 - 2 dot <= tau + eps (|q|^2 + |c|^2)  rearranged so that the epilogue is one fma and
// one compare per pair:   fma(-2, dot, Cn[c]) <= Rq[q],   Cn = (1 - eps) |c|^2,   Rq = tau - (1 - eps) |q|^2.
// (eps carries a factor 2 of slack over the bound above, which also covers these few extra roundings.)
__global__ __launch_bounds__(256) void knn_thresholds_kernel(int64_t n, int64_t q_begin, int64_t nq, int k, float eps,
                                                             const float* __restrict__ norms, const float* __restrict__ sample_d2,
                                                             float* __restrict__ Rq, float* __restrict__ Cn, int32_t* __restrict__ counts,
                                                             uint16_t* __restrict__ A2_fold, int64_t K3, int dp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) Cn[i] = (1.f - eps) * norms[i];
  if (i < nq) {
    const float rq = sample_d2[i * k + (k - 1)] - (1.f - eps) * norms[q_begin + i];  // +inf when the sample held < k points
    Rq[i] = rq;
    counts[i] = 0;
    if (A2_fold) {  // the folded filter: -Rq rides three K columns of the query's A row
      unsigned int r3[3];
      split3(-rq, r3);
      for (int j = 0; j < 3; ++j) A2_fold[(q_begin + i) * K3 + 3 * dp + j] = (uint16_t)r3[j];
    }
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

// d <= 64: "query-stationary" form of the same filter.  The tile kernel above re-reads both operands for every
// 128 x 128 tile — 5.25 bytes of L2 traffic per pair at K3 = 168, which is what bounds it (counters: 4.6 TB/s of L2
// requests, matrix cores 13 % busy).  Here a block of 8 waves owns 256 queries for its whole life: each wave keeps the MFMA
// A-fragments of its 64 queries (all of K3) in REGISTERS, and the block only streams 128-candidate tiles of B2 through a
// double-buffered LDS image (one barrier per tile) — 1.3 bytes per pair.
//
// Round 2: the threshold test itself rides the matrix cores (FOLD layout of knn_split_kernel: the accumulator is
// -2 q.c + Cn[c] - Rq[q], a pair survives iff it is <= 0), so the epilogue is one compare + one add-with-carry per pair
// (was fma + compare against an LDS operand + shift/or), and it is ROTATED half a tile against the MFMAs: while the
// matrix cores work on candidate sub-tile j of tile t the vector ALUs test sub-tile 1-j of the previous half, paired one
// MFMA with a fixed number of test pairs behind it (order written out and pinned).  Measured at 1M x 50 (profiles/): 388 -> 353 ms;
// ablation builds (-DDH_KNN_ABL=1: no appends) run 272 ms, i.e. the survivor appends (~960 per query, one returning LDS
// atomic + one 4-byte global store each) cost 81 ms and the tile loop itself sits at ~59 % matrix-pipe utilisation behind its
// one barrier per 128-candidate tile.  The six extra K columns cost no K-step at d = 50 (3 * 56 + 6 <= 176); their
// products are exact and the extra accumulation roundings (<= 7 u (|q|^2 + |c|^2)) sit inside the factor 2 of slack in eps.
template <int KS>  // 16-wide k steps: K3 = 16 KS
__global__ __launch_bounds__(512) void knn_filter_small_kernel(int64_t nq, int64_t n, const uint16_t* __restrict__ A2,
                                                               const uint16_t* __restrict__ B2, int32_t* __restrict__ counts,
                                                               int32_t* __restrict__ surv, int cap, int seg,
                                                               int64_t tiles_per_slice) {
  constexpr int K3 = 16 * KS;
  constexpr int LD = K3 + 8;               // bf16 per LDS row: (LD / 2) % 8 == 4 -> conflict-free 16-byte fragment reads
  constexpr int CPR = 2 * KS;              // 16-byte chunks per row
  constexpr int NCH = (BN * CPR + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];  // 2 x [BN][LD]
  __shared__ int cnt_s[256];  // survivors appended by THIS block per query: the block owns its 256 queries, so the
                              // list cursors are LDS atomics (~100 cycles) instead of returning global atomics
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 31, kh = (lane >> 5) * 8;
  const int64_t m0 = (int64_t)blockIdx.x * 256 + wr * 64;

  if (tid < 256) cnt_s[tid] = 0;
  bf16x8 a[2][KS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t row = m0 + i * 32 + lr;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      u32x4 v = u32x4(0u);
      if (row < nq) v = *reinterpret_cast<const u32x4*>(A2 + row * K3 + kk * 16 + kh);
      a[i][kk] = __builtin_bit_cast(bf16x8, v);
    }
  }

  int* cnt = cnt_s + wr * 64 + 4 * (lane >> 5);
  const int64_t row0 = m0 + 4 * (lane >> 5);                                      // query of accumulator register (i, r):
  int32_t* surv_base = surv + row0 * cap + (int64_t)blockIdx.y * seg;  // row0 + i * 32 + (r & 3) + 8 * (r >> 2); this slice's segment

  const int64_t tiles_n = (n + BN - 1) / BN;
  const int64_t t_lo = (int64_t)blockIdx.y * tiles_per_slice, t_hi = min(tiles_n, t_lo + tiles_per_slice);
  u32x4 stage[NCH];
  // chunk s of this thread: row c / CPR of the tile, 16-byte column c % CPR (fixed per thread); only the last tile of the
  // matrix can hold rows >= n, so every other tile is loaded without guards (the per-chunk 64-bit compares and branches of
  // the guarded form were a sixth of the kernel's non-MFMA instructions)
  const uint16_t* b_chunk[NCH];
#pragma unroll
  for (int s = 0; s < NCH; ++s) {
    const int c = tid + 512 * s;
    b_chunk[s] = B2 + (int64_t)(c / CPR) * K3 + (c % CPR) * 8;
  }
  const bool chunk_tail_live = tid + 512 * (NCH - 1) < BN * CPR;  // the last chunk index may run past the tile
  auto load_tile = [&](int64_t t) {
    const int64_t off = t * BN * (int64_t)K3;
    if (t * BN + BN <= n) {
#pragma unroll
      for (int s = 0; s < NCH; ++s)
        if (s + 1 < NCH || chunk_tail_live) stage[s] = *reinterpret_cast<const u32x4*>(b_chunk[s] + off);
    } else {
#pragma unroll
      for (int s = 0; s < NCH; ++s) {
        const int c = tid + 512 * s;
        const int64_t row = t * BN + c / CPR;
        stage[s] = (c < BN * CPR && row < n) ? *reinterpret_cast<const u32x4*>(b_chunk[s] + off) : u32x4(0u);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
      const int c = tid + 512 * s;
      if (c < BN * CPR) *reinterpret_cast<u32x4*>(lds + (size_t)buf * BN * LD + (c / CPR) * LD + (c % CPR) * 8) = stage[s];
    }
  };
  f32x16 acc[2][2];
  // Only lanes with a hit walk their set bits (an in-line "if (pass) append" per pair serialised ~10 returning atomics per
  // tile and wave: 65 % of all wave cycles were spent waiting).  Both query sub-tiles advance together, so the two
  // returning LDS atomics of a round are in flight at the same time and the wave waits once per round, not once per hit.
  auto append = [&](int j, int64_t t, const unsigned int (&h)[2]) __attribute__((always_inline)) {
#if defined(DH_KNN_ABL) && (DH_KNN_ABL == 1 || DH_KNN_ABL == 2)
    if (h[0] == 0xdeadbeefu && h[1] == 0x12345u) cnt[0] = 1;  // ablation build: keep h alive, never append
    return;
#endif
    const int64_t c = t * BN + wc * 64 + j * 32 + lr;
    const bool live = c < n;  // zero-padded candidate rows pass the folded test
    unsigned int mk0 = live ? h[0] : 0u, mk1 = live ? h[1] : 0u;
    while (mk0 | mk1) {
      const int b0 = 31 - __clz(mk0 | 1u), b1 = 31 - __clz(mk1 | 1u);  // (| 1: defined for an empty mask, then unused)
      const int r0 = 15 - b0, r1 = 15 - b1;
      const int q0 = (r0 & 3) + 8 * (r0 >> 2), q1 = 32 + (r1 & 3) + 8 * (r1 >> 2);
      // rows beyond the query range hold zero fragments and pass: skip them
      const bool p0 = mk0 != 0u && row0 + q0 < nq, p1 = mk1 != 0u && row0 + q1 < nq;
      int pos0 = 0, pos1 = 0;
      if (p0) pos0 = atomicAdd(cnt + q0, 1);
      if (p1) pos1 = atomicAdd(cnt + q1, 1);
      if (p0 && pos0 < seg) surv_base[(int64_t)q0 * cap + pos0] = (int32_t)c;
      if (p1 && pos1 < seg) surv_base[(int64_t)q1 * cap + pos1] = (int32_t)c;
      mk0 &= ~(1u << b0);
      mk1 &= ~(1u << b1);
    }
  };
  // One half tile: all K steps of candidate sub-tile JM on the matrix cores (fragments four steps ahead of their MFMAs)
  // while the vector ALUs take the pass bits of sub-tile JT from the previous half: bit (15 - r) of h[i] =
  // [acc[i][JT][r] <= 0], one v_cmp + one v_addc (h + h + carry) per pair, a fixed number of pairs behind every MFMA.
  // The order is written out and pinned (sched_barrier): left to itself the compiler sinks the pass bits into the
  // (rarely taken) append branch behind the MFMAs and packs them with three VALU ops per pair.
  auto half_tile = [&](auto jm_tag, const uint16_t* b_frag, unsigned int (&h)[2], int64_t prefetch_tile) __attribute__((always_inline)) {
    constexpr int JM = decltype(jm_tag)::value, JT = 1 - JM;
    constexpr int PER = (32 + 2 * KS - 1) / (2 * KS);  // pairs behind each MFMA
    constexpr int AHEAD = 4;  // fragment reads in flight ahead of their MFMAs (an LDS read is ~128 clocks, an MFMA pair 64)
    bf16x8 b[KS];
#pragma unroll
    for (int kk = 0; kk < AHEAD && kk < KS; ++kk) b[kk] = *reinterpret_cast<const bf16x8*>(b_frag + JM * 32 * LD + kk * 16);
    h[0] = h[1] = 0u;
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
      const int kk = m >> 1, i = m & 1;
      if (i == 0 && kk + AHEAD < KS) b[kk + AHEAD] = *reinterpret_cast<const bf16x8*>(b_frag + JM * 32 * LD + (kk + AHEAD) * 16);
      if (kk == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        acc[i][JM] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], b[kk], z, 0, 0, 0);
      } else {
        acc[i][JM] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], b[kk], acc[i][JM], 0, 0, 0);
      }
#pragma unroll
      for (int e = m * PER; e < (m + 1) * PER && e < 32; ++e)
        asm("v_cmp_ge_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(h[e >> 4]) : "v"(acc[e >> 4][JT][e & 15]) : "vcc");
      __builtin_amdgcn_sched_barrier(0);
      // the next tile's loads go out behind the FIRST MFMA of the tile: the compiler drains vmcnt (the previous half's append
      // stores share the counter with loads) in front of this block, so this is the earliest point that keeps them in flight
      if (JM == 0 && m == 0 && prefetch_tile >= 0) {
        load_tile(prefetch_tile);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  load_tile(t_lo < t_hi ? t_lo : 0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  unsigned int h[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][1][r] = 1.f;  // nothing passes in the first rotated half
  for (int64_t t = t_lo; t < t_hi; ++t) {
    const uint16_t* b_frag = lds + (size_t)cur * BN * LD + (wc * 64 + lr) * LD + kh;
    // half A: matrix cores on sub-tile 0 of tile t, vector ALUs on sub-tile 1 of tile t - 1 (first tile: preset to "no pass")
    __builtin_amdgcn_sched_barrier(0);
    half_tile(std::integral_constant<int, 0>{}, b_frag, h, t + 1 < t_hi ? t + 1 : -1);
    append(1, t - 1, h);
    // half B: matrix cores on sub-tile 1, vector ALUs on sub-tile 0 of the same tile
    __builtin_amdgcn_sched_barrier(0);
    half_tile(std::integral_constant<int, 1>{}, b_frag, h, -1);
    append(0, t, h);
    if (t + 1 < t_hi) store_tile(cur ^ 1);
#if defined(DH_KNN_ABL) && DH_KNN_ABL == 3
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ablation build: no barrier (results are garbage)
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
      for (int r = 0; r < 16; ++r) h[i] = h[i] + h[i] + (unsigned int)(acc[i][1][r] <= 0.f);
    }
    append(1, t_hi - 1, h);
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
    for (int64_t t = 0; t < d; t += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xq + t), b = *reinterpret_cast<const f32x4*>(xc + t);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float diff = __fsub_rn(a[u], b[u]);
        acc = __fadd_rn(acc, __fmul_rn(diff, diff));
      }
    }
  } else {
    for (int64_t t = 0; t < d; ++t) {
      const float diff = __fsub_rn(xq[t], xc[t]);
      acc = __fadd_rn(acc, __fmul_rn(diff, diff));
    }
  }
  return acc;
}

template <bool VEC>
__global__ __launch_bounds__(256) void knn_rerank_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                         int64_t q_begin, int64_t nq, int k, const int32_t* __restrict__ counts,
                                                         const int32_t* __restrict__ surv, int cap, int n_seg, int seg,
                                                         int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 4 + wave;
  if (q >= nq) return;
  // per wavefront: keys[0 .. RR_CHUNK) = this round's candidates, keys[RR_CHUNK .. RR_CHUNK + k) = best so far
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem) + (size_t)wave * (RR_CHUNK + 64);
  const unsigned long long kInf = ~0ull;
  const float* xq = X + (q_begin + q) * ldx;
  // the list of query q is n_seg segments of `seg` slots (one per candidate slice of the filter); counts[s * nq + q]
  // entries of segment s are valid.  Any segment over capacity: the list is incomplete -> every candidate is re-scanned.
  bool overflow = false;
  for (int sg = 0; sg < n_seg; ++sg) overflow |= counts[(int64_t)sg * nq + q] > seg;
  int carried = 0;
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
      out_dist[q * k + s] = (float)sqrt((double)__uint_as_float((unsigned int)(v >> 32)));
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

int knn_filter_cap(int64_t n, int k) {
  // expected survivors ~ k n / S (times the eps margin); 4x that, at least 256, a power of two
  const int64_t want = 4 * (int64_t)k * dh::ceil_div(n, knn_filter_sample_size(n));
  int cap = 256;
  while (cap < want && cap < (1 << 16)) cap <<= 1;
  return cap;  // (split into per-slice segments by knn_filter_launch; an overfull segment only costs that query a re-scan)
}

// candidate slices of the query-stationary filter (each gets its own segment of every survivor list and its own
// counter array): enough blocks for two rounds of the chip
int knn_filter_slices(int64_t nq) {
  const int64_t qblocks = ceil_div(nq, 256);
  return (int)(qblocks >= 512 ? 1 : ceil_div(512, qblocks));
}

// bf16 columns of an operand row: three dp-wide parts, for d <= 64 (the query-stationary kernel) plus the six columns that
// carry the folded thresholds; rounded up to whole 16-wide MFMA steps
bool knn_filter_folds(int64_t d) { return (d + 7) / 8 * 8 <= 64; }
int64_t knn_filter_k3(int64_t d) { return (3 * (int64_t)((d + 7) / 8 * 8) + (knn_filter_folds(d) ? 6 : 0) + 15) / 16 * 16; }

// Survivor lists: n_seg segments of `seg` slots per query (row stride n_seg * seg).  One segment for the tile kernel
// (d > 64); one per candidate slice for the query-stationary kernel, each at least 1024 slots deep so that a query
// whose neighbours all sit in one slice does not overflow (an overfull segment only costs that query a re-scan).
void knn_filter_geometry(int64_t n, int64_t d, int64_t nq, int k, int* n_seg, int* seg) {
  const int cap = knn_filter_cap(n, k);
  *n_seg = 1;
  *seg = cap;
  if (knn_filter_folds(d)) {
    const int64_t tiles_n = ceil_div(n, BN);
    int64_t slices = knn_filter_slices(nq);
    if (slices > tiles_n) slices = tiles_n;
    const int64_t tps = ceil_div(tiles_n, slices);
    *n_seg = (int)ceil_div(tiles_n, tps);
    int sg = cap / *n_seg;
    if (*n_seg > 1 && sg < 1024) sg = cap < 1024 ? cap : 1024;
    *seg = sg;
  }
}

int knn_filter_padded_d(int64_t d) { return (int)((d + 7) / 8 * 8); }


void knn_filter_sample(int64_t n, int64_t d, const float* X, int64_t ldx, int rs, float* Xs, hipStream_t st) {
  const int64_t S = knn_filter_sample_size(n);
  hipLaunchKernelGGL(knn_sample_kernel, dim3((unsigned)ceil_div(S * rs, 256)), dim3(256), 0, st, S, n / S, d, X, ldx, rs, Xs);
}

// Steps 2 and 3 (the sample pass has already left the raw k-th sample distances in sample_d2 [nq][k]).
// `Xr` (leading dimension ldr, dr >= d columns, extra columns zero) is what the re-rank reads: the zero-padded copy for
// d <= 64, X itself otherwise.
int knn_filter_launch(int64_t n, int64_t d, const float* X, int64_t ldx, const float* Xr, int64_t ldr, int64_t dr,
                      int64_t q_begin, int64_t nq, int k,
                      const float* sample_d2, float* mean_ws, uint16_t* A2, uint16_t* B2, float* norms, float* Rq, float* Cn,
                      int32_t* counts, int32_t* surv, int32_t* out_idx, float* out_dist, hipStream_t st) {
  const int dp = knn_filter_padded_d(d);
  const int64_t K3 = knn_filter_k3(d);
  const float eps = 1.220703125e-4f + (float)d * 9.5367431640625e-7f;  // 2^-13 + d 2^-20
  const int n_partial = (int)(n < kMeanBlocks ? n : kMeanBlocks);
  hipLaunchKernelGGL(knn_colsum_kernel, dim3((unsigned)n_partial), dim3(256), 0, st, n, d, X, ldx, mean_ws + d);
  hipLaunchKernelGGL(knn_mean_kernel, dim3((unsigned)ceil_div(d, 256)), dim3(256), 0, st, n, d, n_partial, mean_ws + d, mean_ws);
  const bool fold = knn_filter_folds(d);
  if (fold)
    hipLaunchKernelGGL(knn_split_kernel<true>, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, n, d, X, ldx, mean_ws, dp, K3, eps, A2, B2, norms);
  else
    hipLaunchKernelGGL(knn_split_kernel<false>, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, n, d, X, ldx, mean_ws, dp, K3, eps, A2, B2, norms);
  hipLaunchKernelGGL(knn_thresholds_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, q_begin, nq, k, eps, norms,
                     sample_d2, Rq, Cn, counts, fold ? A2 : nullptr, K3, dp);
  int n_seg, seg;
  knn_filter_geometry(n, d, nq, k, &n_seg, &seg);
  const int cap = n_seg * seg;  // slots per query
  if (fold) {  // query-stationary kernel; candidate tiles sliced over grid.y until >= 2 rounds of blocks exist
    const int64_t qblocks = ceil_div(nq, 256), tiles_n = ceil_div(n, BN);
    const int64_t tps = ceil_div(tiles_n, n_seg);
    dim3 grid((unsigned)qblocks, (unsigned)n_seg);
    const int ks = (int)(K3 / 16);
    const size_t lds = 2 * (size_t)BN * (K3 + 8) * sizeof(uint16_t);
#define DH_KNN_FS(KS)                                                                                                          \
  case KS: {                                                                                                                   \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_filter_small_kernel<KS>),                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,                                     \
                                               (int)(2 * BN * (16 * KS + 8) * sizeof(uint16_t))) == hipSuccess;                \
    if (!ok) return fail(DH_ERR_LAUNCH, "dh_knn_bruteforce_f32: cannot raise the dynamic LDS limit");                        \
    hipLaunchKernelGGL(knn_filter_small_kernel<KS>, grid, dim3(512), lds, st, nq, n, A2 + q_begin * K3, B2, counts,            \
                       surv, cap, seg, tps);                                                                                   \
  } break
    switch (ks) {
      DH_KNN_FS(2); DH_KNN_FS(4); DH_KNN_FS(5); DH_KNN_FS(7); DH_KNN_FS(8); DH_KNN_FS(10); DH_KNN_FS(11); DH_KNN_FS(13);
      default: return fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: unexpected filter depth %d", ks);
    }
#undef DH_KNN_FS
  } else {
    const int64_t tiles = ceil_div(nq, BM) * ceil_div(n, BN);
    if (tiles >= (int64_t)1 << 31) return fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: too many filter tiles (%lld)", (long long)tiles);
    constexpr size_t kTileLds = (size_t)TILE_LDS_ELEMS * sizeof(uint16_t);
    static const bool tile_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)kTileLds) == hipSuccess;
    if (!tile_ok) return fail(DH_ERR_LAUNCH, "dh_knn_bruteforce_f32: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(knn_filter_kernel, dim3((unsigned)tiles), dim3(256), kTileLds, st, nq, n, K3, A2 + q_begin * K3, B2, Rq, Cn, counts,
                       surv, cap);
  }
  const size_t lds = 4 * (size_t)(RR_CHUNK + 64) * sizeof(unsigned long long);
  const bool vec = dr % 4 == 0 && ldr % 4 == 0 && aligned16(Xr);
  if (vec)
    hipLaunchKernelGGL(knn_rerank_kernel<true>, dim3((unsigned)ceil_div(nq, 4)), dim3(256), lds, st, n, dr, Xr, ldr, q_begin, nq, k, counts,
                       surv, cap, n_seg, seg, out_idx, out_dist);
  else
    hipLaunchKernelGGL(knn_rerank_kernel<false>, dim3((unsigned)ceil_div(nq, 4)), dim3(256), lds, st, n, dr, Xr, ldr, q_begin, nq, k, counts,
                       surv, cap, n_seg, seg, out_idx, out_dist);
  return check_launch("dh_knn_bruteforce_f32(filter)");
}

}  // namespace dh
