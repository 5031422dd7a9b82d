// Dense feature GEMM on the CDNA4 f32 matrix cores.
//
// C[M,N] (+)= op(A)[M,K] * op(B)[K,N], exact f32: v_mfma_f32_32x32x2_f32 is bit-for-bit a
// k-ordered fmaf chain (one rounding per product), so the layer keeps the reference's fp32
// semantics (torch.mm, scdsc.py:497 / spagcn.py:358) — gfx950 has no TF32-like mode and we
// do not down-convert.  MFMA-bound: 157 TFLOP/s peak for f32 inputs.
//
// Block = 8 wavefronts computing a 256x256 tile; wave (wm, wn) of the 2x4 wave grid owns a
// 128x64 patch = 4x2 MFMA tiles (128 accumulator VGPRs).  The big macro-tile is deliberate:
// 256x256 halves the bytes fetched per flop relative to 128x128.  K is consumed 32 at a time through a register-staged,
// double-buffered LDS pipeline two k-groups deep (see the main loop), one barrier per K-step, and inside each k-group every
// LDS read / LDS write / global load is paired with one MFMA by sched_group_barrier, so a wave's non-MFMA issue slots hide
// behind its own 64-clock MFMAs.  What was measured on the way (1M x 2000 x 512, TFLOP/s NN / TN, rocBLAS 148 / 137-142):
//   bursts of memory instructions between 32-MFMA blocks, refill one k-group deep      136.7 / 137.0
//   + all tiles of a K-slice on one XCD (TN traffic 25.4 -> 10.05 GB = algorithmic)     136.7 / 137.9
//   + refill two k-groups deep, no vmcnt(0) at the barrier (bursts kept)                135.7 / 138.2   (latency was not it)
//   + one memory instruction per MFMA                                                   139.8 / 142.7
//   + constant descriptors, K advance on the lane offsets                               141.8 / 143.8
// The 128x128 configuration (2 blocks per CU) runs the same loop at 140.1 / 141.0.
// LDS images are padded so fragment reads are bank-conflict free:
//   "MK" image (operand stored with K contiguous): [256][32+4] floats; a fragment is one
//        ds_read_b128 per 32x32 tile (4 consecutive k of one row) feeding 4 MFMA steps; the
//        wave's tiles are consecutive 32-row slabs of its span;
//   "KM" image (operand stored with M/N contiguous): [32][256+4] floats; a fragment is one
//        ds_read_b128 (A side, 4 ADJACENT rows) or ds_read_b64 (B side, 2 adjacent columns)
//        per MFMA step feeding ALL of the wave's tiles on that side, which therefore
//        interleave: tile x holds rows 4i + x (columns 2j + y) of the wave's span.
// Lane half h = lane>>5 supplies k = 8*g + 4*h + s at MFMA step s of k-group g for BOTH
// operands, so the hardware's (k = lane>>5) pairing is a permutation of the K-slice.
// Fragments of k-group g+1 are fetched from LDS while the 32 MFMAs of group g execute, and the
// per-K-step barrier sits in front of the LAST k-group so that it, the LDS store->load turnaround
// and the next tile's first fragment reads hide behind that group's MFMAs.  Problems with too few
// 256x256 tiles to fill the chip use the same kernel at 128x128 (4 waves, 2x2 tiles each).
//
// The transposed-A form (dW = X^T dZ, K = number of cells) is split over K into slices of a linear grid;
// partial slabs are summed by a second deterministic kernel (no float atomics).  Block ids are remapped so
// that consecutive tiles (one K-slice: same A row panel; split-K: the whole slice) land on the same XCD (private L2).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A/B switches for scripts/gemm_variants.sh (defaults = the measured winners)
#ifndef DH_GEMM_LIVEGROUPS
#define DH_GEMM_LIVEGROUPS 1
#endif
#ifndef DH_GEMM_SLICEMAP
#define DH_GEMM_SLICEMAP 1  // split-K: all tiles of a K-slice on one XCD
#endif

constexpr int BK = 32;
constexpr int LD_MK = BK + 4;  // K-contiguous image: 36-float rows (16-B aligned, b128 reads conflict-free)

// Geometry of one kernel configuration: WM x WN wavefronts, each owning TM x TN MFMA tiles of 32x32.
template <int WM, int WN, int TM_, int TN_>
struct Cfg {
  static constexpr int TM = TM_, TN = TN_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  static constexpr int WAVES_N = WN;
  static constexpr int NLD_A = BM * BK / 4 / NT, NLD_B = BN * BK / 4 / NT;  // float4 pieces per thread per tile
  static constexpr int TILE_A = (BM * LD_MK > BK * (BM + 4)) ? BM * LD_MK : BK * (BM + 4);
  static constexpr int TILE_B = (BN * LD_MK > BK * (BN + 4)) ? BN * LD_MK : BK * (BN + 4);
  static_assert(NLD_A % 2 == 0 && NLD_B % 2 == 0, "refill is issued in two halves");
};
using CfgLarge = Cfg<2, 4, 4, 2>;  // 256 x 256, 512 threads, 1 block / CU
using CfgSmall = Cfg<2, 2, 2, 2>;  // 128 x 128, 256 threads, 2 blocks / CU (few-tile problems)

// One float4 piece of an operand tile: global -> register.  KCONTIG: operand stored [rows][K]
// (k contiguous); otherwise stored [K][rows].  ROWS_T = tile extent (BM or BN), NT = block threads.
template <bool KCONTIG, bool ALIGNED, int ROWS_T, int NT>
__device__ __forceinline__ f32x4 load_piece(int r, const float* __restrict__ P, int64_t ld, int64_t rows, int64_t r0,
                                            int64_t k0, int64_t k_end, int tid) {
  const int idx = tid + NT * r;
  f32x4 v = f32x4(0.f);
  if constexpr (KCONTIG) {
    const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
    const int64_t gr = min(r0 + row, rows - 1);  // clamped rows are never stored
    const int64_t gk = k0 + kq;
    const float* p = P + gr * ld + gk;
    if constexpr (ALIGNED) {
      if (gk < k_end) v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (gk + i < k_end) v[i] = p[i];
    }
  } else {
    const int kk = idx / (ROWS_T / 4), mq = (idx % (ROWS_T / 4)) * 4;
    const int64_t gk = k0 + kk;
    const int64_t gm = r0 + mq;
    if (gk < k_end) {
      const float* p = P + gk * ld + gm;
      if (ALIGNED && gm + 3 < rows) v = *reinterpret_cast<const f32x4*>(p);
      else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (gm + i < rows) v[i] = p[i];
      }
    }
  }
  return v;
}

template <bool KCONTIG, int ROWS_T, int NT>
__device__ __forceinline__ void store_piece(int r, float* __restrict__ lds, f32x4 v, int tid) {
  const int idx = tid + NT * r;
  if constexpr (KCONTIG) {
    const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
    *reinterpret_cast<f32x4*>(lds + row * LD_MK + kq) = v;
  } else {
    const int kk = idx / (ROWS_T / 4), mq = (idx % (ROWS_T / 4)) * 4;
    *reinterpret_cast<f32x4*>(lds + kk * (ROWS_T + 4) + mq) = v;
  }
}

// Fragments of one operand side for k-group g: f[x][s] = value of the wave's tile x at MFMA step s.
// Plain float arrays (not vectors): the KM image delivers the tiles' values transposed, and scalar
// registers let the compiler feed the ds_read results to the MFMAs in place (no v_mov shuffles).
template <bool KCONTIG, int T, int ROWS_T>
__device__ __forceinline__ void read_frags(float (&f)[T][4], const float* __restrict__ lds, int span0, int i32, int g, int h) {
  if constexpr (KCONTIG) {
#pragma unroll
    for (int x = 0; x < T; ++x) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + (span0 + x * 32 + i32) * LD_MK + g * 8 + h * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s) f[x][s] = v[s];
    }
  } else {
    const float* p = lds + (g * 8 + h * 4) * (ROWS_T + 4) + span0 + T * i32;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (T == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + s * (ROWS_T + 4));
#pragma unroll
        for (int x = 0; x < 4; ++x) f[x][s] = v[x];
      } else {
        static_assert(T == 2, "2 or 4 tiles per side");
        const f32x2 v = *reinterpret_cast<const f32x2*>(p + s * (ROWS_T + 4));
        f[0][s] = v[0];
        f[1][s] = v[1];
      }
    }
  }
}

// Byte offset of piece r of this thread inside an operand tile whose origin is (row r0, k = 0); loop invariant.
// Rows beyond the matrix are clamped (their results are never stored).
template <bool KCONTIG, int ROWS_T, int NT>
__device__ __forceinline__ uint32_t piece_offset(int r, int64_t ld, int64_t rows, int64_t r0, int tid) {
  const int idx = tid + NT * r;
  if constexpr (KCONTIG) {
    const int row = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
    const int64_t rel = min((int64_t)row, rows - 1 - r0);
    return (uint32_t)((rel * ld + kq) * 4);
  } else {
    const int kk = idx / (ROWS_T / 4), mq = (idx % (ROWS_T / 4)) * 4;
    return (uint32_t)(((int64_t)kk * ld + mq) * 4);
  }
}

// k index (inside the tile) of piece r of this thread
template <bool KCONTIG, int ROWS_T, int NT>
__device__ __forceinline__ int piece_k(int r, int tid) {
  const int idx = tid + NT * r;
  return KCONTIG ? (idx % (BK / 4)) * 4 : idx / (ROWS_T / 4);
}

// Buffer descriptor over [base, base + 4 GiB) built from a wave-uniform pointer (readfirstlane makes the
// uniformity provable, so no waterfall loop is generated around the loads).
// num_records = bytes up to the end of the matrix (capped at 4 GiB): loads past it return 0 instead of faulting.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, const char* end) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const int64_t left = end - base;
  const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(left < 0 ? 0 : (left > 0xffffffffLL ? 0xffffffffLL : left)));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 buffer_load_x4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

// position inside the wave's span of MFMA index i (0..31) of tile x (T tiles on this side)
template <bool KCONTIG, int T>
__device__ __forceinline__ int span_pos(int x, int i) { return KCONTIG ? x * 32 + i : T * i + x; }

// TA: A stored [K][M]; TB: B stored [N][K].
template <typename C_, bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(C_::NT, 2) void gemm_f32_kernel(
    int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int accumulate, int64_t k_chunk, float* __restrict__ slabs, int tiles_n, int n_tiles, int n_slices,
    const float* __restrict__ bias, int act) {
  constexpr int TM = C_::TM, TN = C_::TN, BM = C_::BM, BN = C_::BN, NT = C_::NT;
  constexpr int NLA = C_::NLD_A, NLB = C_::NLD_B;
  __shared__ __attribute__((aligned(16))) float lds[2 * C_::TILE_A + 2 * C_::TILE_B];  // A[0], A[1], B[0], B[1]
  float* const lds_a = lds;
  float* const lds_b = lds + 2 * C_::TILE_A;

  // XCD-aware bijective remap of the LINEAR block id b (the dispatcher places block b on XCD b % 8, observed):
  //  * one K-slice: every XCD gets a contiguous run of logical tiles, so neighbours (same A row panel) share one L2;
  //  * split-K (dW = X^T dS): all tiles of a K-slice run on ONE XCD at the same time (slice = xcd + 8 * round), so the
  //    slice's rows of X and dS are fetched from HBM once and the other tiles of the slice hit that XCD's L2 — the
  //    row tiles re-reading dS (and the column tiles re-reading X) were 2.5x the algorithmic traffic before.
#ifdef DH_GEMM_SETPRIO  // A/B switch (scripts/overlap_diag.sh): wave priority of the GEMM next to co-resident aggregation waves
  __builtin_amdgcn_s_setprio(DH_GEMM_SETPRIO);
#endif
  const int bid = blockIdx.x;
  int logical, slice;
  if (n_slices == 1) {
    const int q = n_tiles / 8, rr = n_tiles % 8, xcd = bid % 8;
    logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + bid / 8;
    slice = 0;
  } else if (DH_GEMM_SLICEMAP && n_slices % 8 == 0) {
    const int xcd = bid % 8, j = bid / 8;
    slice = xcd + 8 * (j / n_tiles);
    logical = j % n_tiles;
  } else {
    logical = bid % n_tiles;
    slice = bid / n_tiles;
  }
  const int64_t m0 = (int64_t)(logical / tiles_n) * BM;
  const int64_t n0 = (int64_t)(logical % tiles_n) * BN;

  const int64_t k_begin = (int64_t)slice * k_chunk;
  const int64_t k_end = min(K, k_begin + k_chunk);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C_::WAVES_N, wn = wave % C_::WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  const int a_span = wm * (TM * 32), b_span = wn * (TN * 32);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x16(0.f);

  const int64_t n_steps = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;
  float fa[2][TM][4], fb[2][TN][4];  // fragments [register set][tile][MFMA step]
  if (n_steps > 0) {
#pragma unroll
    for (int r = 0; r < NLA; ++r)
      store_piece<!TA, BM, NT>(r, lds_a, load_piece<!TA, ALIGNED, BM, NT>(r, A, lda, M, m0, k_begin, k_end, tid), tid);
#pragma unroll
    for (int r = 0; r < NLB; ++r)
      store_piece<TB, BN, NT>(r, lds_b, load_piece<TB, ALIGNED, BN, NT>(r, B, ldb, N, n0, k_begin, k_end, tid), tid);
  }
  __syncthreads();
  if (n_steps > 0) {
    read_frags<!TA, TM, BM>(fa[0], lds_a, a_span, i32, 0, h);
    read_frags<TB, TN, BN>(fb[0], lds_b, b_span, i32, 0, h);
  }

  // Fast refill path: when a whole tile lies inside the matrix (always, except the K tail) a piece is ONE unguarded
  // 16-byte buffer load: one descriptor per operand per block + a 32-bit lane offset.  Every non-MFMA instruction in the
  // main loop displaces matrix-pipe time (measured: the guarded, 64-bit-addressed refill cost 17 % of the kernel), so the
  // main loop contains only the fast path and the guarded path runs in a separate tail loop.
  uint32_t offa[NLA], offb[NLB];
#pragma unroll
  for (int r = 0; r < NLA; ++r) offa[r] = piece_offset<!TA, BM, NT>(r, lda, M, m0, tid);
#pragma unroll
  for (int r = 0; r < NLB; ++r) offb[r] = piece_offset<TB, BN, NT>(r, ldb, N, n0, tid);
  // Edge tiles take the fast path too: K-contiguous operands clamp their rows; M/N-contiguous operands read up
  // to 3 floats past column M (N) of a row, i.e. into the next row or (last row) past the matrix end, where the
  // buffer descriptor returns 0 — such columns only feed C rows/columns >= M (N), which are never stored.
  const bool tile_inside = ALIGNED;
  const char* const a_end = reinterpret_cast<const char*>(A + (TA ? (K - 1) * lda + M : (M - 1) * lda + K));
  const char* const b_end = reinterpret_cast<const char*>(B + (TB ? (N - 1) * ldb + K : (K - 1) * ldb + N));
  const char* const a_origin = reinterpret_cast<const char*>(A + (TA ? m0 : m0 * lda));
  const char* const b_origin = reinterpret_cast<const char*>(B + (TB ? n0 * ldb : n0));
  const int64_t a_kstride = (TA ? lda : 1) * 4, b_kstride = (TB ? 1 : ldb) * 4;  // bytes per unit of k
  // Refill pipeline, two k-groups deep.  The 16-byte pieces of a tile are split in two halves that share one set of
  // staging registers (pa, pb), live across K-steps:
  //   g = 0: retire half 0 of tile t+1 into the idle LDS buffer, issue half 1 of tile t+1, frags g=1
  //   g = 1: frags g=2
  //   g = 2: retire half 1 of tile t+1, issue half 0 of tile t+2, frags g=3
  //   g = 3: BARRIER, frags g=0 of tile t+1 from the freshly filled buffer
  // Every global load has two k-groups (64 MFMAs, ~4000 clocks per wave) between issue and the ds_write that consumes it,
  // and the barrier carries no vmcnt(0): the loads of tile t+2 stay in flight across it.
  // Steps whose tiles t+1 AND t+2 are completely inside K take the unguarded fast path: (t + 3) * BK <= k_len.
  // One descriptor per operand per block (origin at k_begin, records up to the end of the matrix); a lane's offset is
  // piece offset + k * k-stride and advances by one K-step after each use (8 VALU adds per K-step; rebuilding four
  // descriptors per step cost ~60 scalar / 64-bit compare instructions: 139.8 -> 141.8 TFLOP/s NN).  Needs the whole
  // K-slice within 32 bits of the origin
  // (the host checks that and runs the unaligned instantiation otherwise: fast_path_ok)
  const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(a_origin + k_begin * a_kstride, a_end);
  const __amdgpu_buffer_rsrc_t rb0 = make_rsrc(b_origin + k_begin * b_kstride, b_end);
  const uint32_t kstep_a = (uint32_t)(BK * a_kstride), kstep_b = (uint32_t)(BK * b_kstride);
#pragma unroll
  for (int r = 0; r < NLA; ++r) offa[r] += (r < NLA / 2 ? 2u : 1u) * kstep_a;  // half 0 is first loaded for tile 2, half 1 for tile 1
#pragma unroll
  for (int r = 0; r < NLB; ++r) offb[r] += (r < NLB / 2 ? 2u : 1u) * kstep_b;
  const int64_t n_fast = tile_inside ? max((int64_t)0, min(n_steps, (k_end - k_begin) / BK - 2)) : 0;
  f32x4 pa[NLA / 2], pb[NLB / 2];  // refill pieces in flight
  // MODE 1: unguarded buffer loads (tile completely inside the K-slice); MODE 2: the same loads, pieces whose k lies beyond
  // the slice are zeroed on their way into LDS (the K tail: a piece is 4 consecutive k of one row or 4 columns of one k, and
  // K % 4 == 0 on this path, so a piece is entirely in or out); MODE 0: guarded 64-bit-addressed loads (unaligned operands).
  auto issue_half = [&](int half, int64_t k0, auto mode_tag) __attribute__((always_inline)) {
    if constexpr (decltype(mode_tag)::value != 0) {
#pragma unroll
      for (int r = 0; r < NLA / 2; ++r) {
        pa[r] = buffer_load_x4(ra0, offa[half * (NLA / 2) + r]);
        offa[half * (NLA / 2) + r] += kstep_a;
      }
#pragma unroll
      for (int r = 0; r < NLB / 2; ++r) {
        pb[r] = buffer_load_x4(rb0, offb[half * (NLB / 2) + r]);
        offb[half * (NLB / 2) + r] += kstep_b;
      }
    } else {
#pragma unroll
      for (int r = 0; r < NLA / 2; ++r)
        pa[r] = load_piece<!TA, ALIGNED, BM, NT>(half * (NLA / 2) + r, A, lda, M, m0, k0, k_end, tid);
#pragma unroll
      for (int r = 0; r < NLB / 2; ++r)
        pb[r] = load_piece<TB, ALIGNED, BN, NT>(half * (NLB / 2) + r, B, ldb, N, n0, k0, k_end, tid);
    }
  };
  auto retire_half = [&](int half, float* a_nxt, float* b_nxt, int64_t k0, auto mode_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mode_tag)::value == 2;
#pragma unroll
    for (int r = 0; r < NLA / 2; ++r) {
      f32x4 v = pa[r];
      if (MASK && k0 + piece_k<!TA, BM, NT>(half * (NLA / 2) + r, tid) >= k_end) v = f32x4(0.f);
      store_piece<!TA, BM, NT>(half * (NLA / 2) + r, a_nxt, v, tid);
    }
#pragma unroll
    for (int r = 0; r < NLB / 2; ++r) {
      f32x4 v = pb[r];
      if (MASK && k0 + piece_k<TB, BN, NT>(half * (NLB / 2) + r, tid) >= k_end) v = f32x4(0.f);
      store_piece<TB, BN, NT>(half * (NLB / 2) + r, b_nxt, v, tid);
    }
  };
  using Slow = std::integral_constant<int, 0>;
  using Fast = std::integral_constant<int, 1>;
  using FastTail = std::integral_constant<int, 2>;
  if (n_steps > 1) issue_half(0, k_begin + BK, Slow{});

  auto k_step = [&](int64_t t, auto mode_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(mode_tag)::value == 1;
    const int cur = t & 1;
    const bool has1 = FAST || (t + 1 < n_steps), has2 = FAST || (t + 2 < n_steps);
    const int64_t k1 = k_begin + (t + 1) * BK, k2 = k1 + BK;
    const float* a_lds = lds_a + cur * C_::TILE_A;
    const float* b_lds = lds_b + cur * C_::TILE_B;
    float* a_nxt = lds_a + (cur ^ 1) * C_::TILE_A;  // last read at g=2 of step t-1, before that step's barrier
    float* b_nxt = lds_b + (cur ^ 1) * C_::TILE_B;
    // One scheduling region per k-group; inside it every memory instruction is paired with one MFMA (64 clocks of matrix
    // pipe each), so the wave's non-MFMA issue slots hide behind its own MFMAs instead of forming a burst during which the
    // pipe only runs if the SIMD's other wave happens to be in a different phase (both run this same code in step).
    // the last step of a K-slice runs only the k-groups that hold data (K = 2000: 16 of its 32 k)
    const int live_groups = FAST ? BK / 8 : (int)((min(k_end - (k1 - BK), (int64_t)BK) + 7) / 8);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      if (DH_GEMM_LIVEGROUPS && !FAST && g >= live_groups) break;  // only ever true in the last step (has1 == false): nothing left to retire
      if (g == BK / 8 - 1 && has1) {
        // this wave's ds_writes of tile t+1 are complete (lgkmcnt), then the workgroup meets; no vmcnt(0): the loads of
        // tile t+2 stay in flight.  Nobody reads tile t's buffer any more (the g=3 fragments are in registers).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      // source order = LDS order the compiler must keep (it cannot prove the two LDS buffers distinct): fragment reads of
      // the live buffer first, then the writes into the idle one, then the loads that reuse the staging registers
      if (g + 1 < BK / 8) {
        read_frags<!TA, TM, BM>(fa[(g + 1) & 1], a_lds, a_span, i32, g + 1, h);
        read_frags<TB, TN, BN>(fb[(g + 1) & 1], b_lds, b_span, i32, g + 1, h);
      } else if (has1) {
        read_frags<!TA, TM, BM>(fa[0], a_nxt, a_span, i32, 0, h);
        read_frags<TB, TN, BN>(fb[0], b_nxt, b_span, i32, 0, h);
      }
      if (g == 0 && has1) {
        retire_half(0, a_nxt, b_nxt, k1, mode_tag);
        issue_half(1, k1, mode_tag);
      }
      if (g == 2) {
        if (has1) retire_half(1, a_nxt, b_nxt, k1, mode_tag);
        if (has2) issue_half(0, k2, mode_tag);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
          for (int y = 0; y < TN; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][x][s], fb[g & 1][y][s], acc[x][y], 0, 0, 0);
      // issue order of the region: (MFMA, LDS read) x 8, then for g = 0, 2 (MFMA, LDS write) x 4, (MFMA, global load) x 4,
      // then the remaining MFMAs; groups that find fewer candidates than asked for stay short
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (g == 0 || g == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * 4 - 16, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * 4 - 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int64_t t = 0;
  for (; t < n_fast; ++t) k_step(t, Fast{});
  if constexpr (ALIGNED) {
    for (; t < n_steps; ++t) k_step(t, FastTail{});
  } else {
    for (; t < n_steps; ++t) k_step(t, Slow{});
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
  // Epilogue: C (+)= acc (+ bias[col], then ReLU): torch.nn.Linear's bias and a following ReLU in the store that writes the tile —
  // the separate dh_bias_act_f32 pass read and wrote every output once more (19.9 of scDSC's 343 ms epoch at 1M cells were that
  // pass over the autoencoder's activations).  Same arithmetic, same rounding: acc + b, then max(., 0).  Split-K partial slabs carry
  // neither; the reduce kernel applies both.
  float* out = C;
  int64_t ldo = ldc;
  bool add = accumulate != 0;
  const float* eb = bias;
  bool relu = act == DH_ACT_RELU;
  if (slabs) {
    out = slabs + (int64_t)slice * M * N;
    ldo = N;
    add = false;
    eb = nullptr;
    relu = false;
  }
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + a_span + span_pos<!TA, TM>(x, (r & 3) + 8 * (r >> 2) + 4 * h);
      if (row >= M) continue;
      if constexpr (!TB) {
        // B image is N-contiguous: the wave's two column tiles interleave, lane j owns columns 2j, 2j+1
        static_assert(TN == 2, "float2 epilogue assumes two column tiles");
        const int64_t col = n0 + b_span + 2 * i32;
        float* p = out + row * ldo + col;
        const float b0 = (eb && col < N) ? eb[col] : 0.f, b1 = (eb && col + 1 < N) ? eb[col + 1] : 0.f;
        if (col + 1 < N && (ldo % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 7u) == 0)) {
          f32x2 v = {acc[x][0][r], acc[x][1][r]};
          if (add) { const f32x2 o = *reinterpret_cast<const f32x2*>(p); v[0] += o[0]; v[1] += o[1]; }
          if (eb) { v[0] += b0; v[1] += b1; }
          if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
          *reinterpret_cast<f32x2*>(p) = v;
        } else {
          if (col < N) {
            float v = add ? p[0] + acc[x][0][r] : acc[x][0][r];
            if (eb) v += b0;
            p[0] = relu ? fmaxf(v, 0.f) : v;
          }
          if (col + 1 < N) {
            float v = add ? p[1] + acc[x][1][r] : acc[x][1][r];
            if (eb) v += b1;
            p[1] = relu ? fmaxf(v, 0.f) : v;
          }
        }
      } else {
#pragma unroll
        for (int y = 0; y < TN; ++y) {
          const int64_t col = n0 + b_span + y * 32 + i32;
          if (col >= N) continue;
          float* p = out + row * ldo + col;
          float v = add ? (*p + acc[x][y][r]) : acc[x][y][r];
          if (eb) v += eb[col];
          *p = relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
}

// C = (accumulate ? C : 0) + sum_z slabs[z]   (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t M, int64_t N, int S,
                                                            const float* __restrict__ slabs,
                                                            float* __restrict__ C, int64_t ldc,
                                                            int accumulate, const float* __restrict__ bias, int act) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / N, col = i % N;
    float* p = C + row * ldc + col;
    float s = accumulate ? *p : 0.f;
    // eight slab values requested before the first is added (same order of additions): written as one loop this was S dependent round
    // trips per element — 22 us for the 128 slabs of a 50 x 200 weight gradient (graph-sc's large batches: 5.4 ms of a 157 ms epoch in 246
    // of these launches, profiles/r06z_graphsc_epoch_kernels_1M_b8192.md)
    int z = 0;
    for (; z + 8 <= S; z += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(int64_t)(z + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < S; ++z) s += slabs[(int64_t)z * total + i];
    if (bias) s += bias[col];
    *p = act == DH_ACT_RELU ? fmaxf(s, 0.f) : s;
  }
}

struct Plan {
  bool large;
  int tiles_m, tiles_n, n_tiles, S;
  int64_t k_chunk;
};

Plan plan_for(int64_t M, int64_t N, int64_t K, int bm, int bn, bool rows_reduced) {
  Plan p;
  p.large = bm == CfgLarge::BM;
  p.tiles_m = (int)dh::ceil_div(M, bm);
  p.tiles_n = (int)dh::ceil_div(N, bn);
  p.n_tiles = p.tiles_m * p.tiles_n;
  p.S = 1;
  p.k_chunk = dh::ceil_div(K > 0 ? K : 1, BK) * BK;
  // Few output tiles and a long K (dW = X^T dZ): split K so that the grid is (just under) a whole number of
  // rounds of the 256 CUs — floor(1024 / tiles) slices: at most 4 full rounds, the last one >= 98 % full —
  // keeping at least 64 K-steps per block.  Measured at 2000 x 512 x 1M: 1024 blocks 138 TFLOP/s, 4096 blocks 135.
  int64_t S = 1;
  if (p.n_tiles < 512 && K >= 4096) {
    int64_t want = 1024 / p.n_tiles;
    if (want < 1) want = 1;
    int64_t max_s = K / (64 * BK);
    if (max_s < 1) max_s = 1;
    S = want < max_s ? want : max_s;
  }
  // A handful of tiles and a K of a few thousand rows (the dW of a mini-batch: 50 x 200 from 10 k rows) left most of the chip idle
  // under the 64-step rule (8 blocks, 0.25 ms for 0.2 GFLOP at batch 8192 of graph-sc): when the grid would not even cover the
  // 256 CUs once, slices go down to 2 K-steps.  Grids that already fill the chip keep their plan.  (One CU needs 1.7 us per 128 x 128
  // x 32 step of exact-fp32 MFMAs: the 500 x 400 -> 200 Linear of a scDeepSort batch was 8 workgroups x 13 steps = 28 us, launch excluded.)
  // Only for products that reduce over ROWS (trans_a: dW = X^T dZ): their summation order depends on the number of rows anyway.  A
  // forward product (K = features) must not change its K order with M — the layer's outputs are bit-identical on 1 and P GPUs
  // because every rank runs the same K order on its rows (tests/test_gpu_sharded_one_gpu.py).
  if (rows_reduced && p.n_tiles * S < 256 && K >= 256) {
    int64_t want = dh::ceil_div((int64_t)256, (int64_t)p.n_tiles);
    int64_t max_s = K / (2 * BK);
    if (max_s < 1) max_s = 1;
    const int64_t s2 = want < max_s ? want : max_s;
    if (s2 > S) S = s2;
  }
  if (S > 1) {
    p.k_chunk = dh::ceil_div(dh::ceil_div(K, S), BK) * BK;
    p.S = (int)dh::ceil_div(K, p.k_chunk);
  }
  return p;
}

// The 256x256 configuration halves refill traffic per flop but needs >= 2 blocks per CU's worth of work to
// fill the chip; smaller problems use 128x128 tiles.
// tile: DH_GEMM_TILE_AUTO, or a request for one configuration (dh_gemm_f32_ex).  AUTO also avoids 256-wide tiles that
// would be at most half full in N or in M (a 128-column slice of a layer, dh_gcn_layer_*: the large tile computed
// twice the flops there).
Plan make_plan(int64_t M, int64_t N, int64_t K, int tile, bool rows_reduced) {
  Plan big = plan_for(M, N, K, CfgLarge::BM, CfgLarge::BN, rows_reduced);
#ifdef DH_GEMM_FORCE_SMALL
  return plan_for(M, N, K, CfgSmall::BM, CfgSmall::BN, rows_reduced);
#endif
  if (tile == DH_GEMM_TILE_128) return plan_for(M, N, K, CfgSmall::BM, CfgSmall::BN, rows_reduced);
  if (tile == DH_GEMM_TILE_256) return big;
  const bool half_empty = (N % CfgLarge::BN != 0 && N % CfgLarge::BN <= CfgSmall::BN && N < 4 * CfgLarge::BN) ||
                          (M % CfgLarge::BM != 0 && M % CfgLarge::BM <= CfgSmall::BM && M < 4 * CfgLarge::BM);
  Plan small = plan_for(M, N, K, CfgSmall::BM, CfgSmall::BN, rows_reduced);
  if ((int64_t)big.n_tiles * big.S < 512 || half_empty) return small;
  // Both fill the chip: take the one with fewer matrix-core ROUNDS.  A CU runs one 256 x 256 workgroup or two 128 x 128 ones, so a
  // round costs (tile area x K-chunk) x 1 resp. x 2; the last round of a grid is a whole round however few tiles it holds.  At
  // 100k x 2000 x 512 (BASELINE config 2) the 782 large tiles are 3.05 rounds = 4, the 3128 small ones 6.1 = 7 half-cost rounds:
  // 1.91 -> 1.7 ms; at 1M rows both are 31 rounds and the large tile's ~1 % better loop wins (14.25 vs 14.39 ms).  The K order of
  // an output element is the same in both configurations, so a forward product's bits do not depend on the choice.
  const double cost_big = (double)dh::ceil_div((int64_t)big.n_tiles * big.S, (int64_t)256) * (double)big.k_chunk * 4.0;
  const double cost_small = (double)dh::ceil_div((int64_t)small.n_tiles * small.S, (int64_t)512) * (double)small.k_chunk * 2.0 * 1.01;
  return cost_small < cost_big ? small : big;
}

}  // namespace

extern "C" size_t dh_gemm_f32_ex_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, int tile) {
  (void)trans_b;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (tile == DH_GEMM_TILE_AUTO && dh::skinny_applies(M, N, K, trans_a)) return 0;
  Plan p = make_plan(M, N, K, tile, trans_a != 0);
  return p.S > 1 ? (size_t)p.S * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" size_t dh_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b) {
  return dh_gemm_f32_ex_workspace_bytes(M, N, K, trans_a, trans_b, DH_GEMM_TILE_AUTO);
}

extern "C" int dh_gemm_f32(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                           int64_t ldc, int accumulate, void* workspace, size_t workspace_bytes,
                           dh_stream_t stream) {
  return dh_gemm_f32_ex(M, N, K, trans_a, trans_b, A, lda, B, ldb, C, ldc, accumulate, workspace, workspace_bytes, DH_GEMM_TILE_AUTO, stream);
}

namespace {
int gemm_f32_impl(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                  int64_t ldc, int accumulate, const float* bias, int act, void* workspace, size_t workspace_bytes, int tile, dh_stream_t stream);
}

extern "C" int dh_gemm_f32_ex(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                              const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                              int64_t ldc, int accumulate, void* workspace, size_t workspace_bytes,
                              int tile, dh_stream_t stream) {
  return gemm_f32_impl(M, N, K, trans_a, trans_b, A, lda, B, ldb, C, ldc, accumulate, nullptr, DH_ACT_NONE, workspace, workspace_bytes, tile, stream);
}

// C = act(op(A) op(B) + bias): the product with torch.nn.Linear's bias (one value per column of C) and an optional ReLU applied in the
// store of the output tile (split-K: in the reduce kernel; narrow layers: dh_bias_act_f32 behind the streaming kernel).
extern "C" int dh_gemm_f32_bias_act(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B,
                                    int64_t ldb, float* C, int64_t ldc, const float* bias, int act, void* workspace, size_t workspace_bytes,
                                    dh_stream_t stream) {
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32_bias_act: bad act %d", act);
  return gemm_f32_impl(M, N, K, trans_a, trans_b, A, lda, B, ldb, C, ldc, 0, bias, act, workspace, workspace_bytes, DH_GEMM_TILE_AUTO, stream);
}

namespace {
int gemm_f32_impl(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                  int64_t ldc, int accumulate, const float* bias, int act, void* workspace, size_t workspace_bytes, int tile, dh_stream_t stream) {
  if (tile != DH_GEMM_TILE_AUTO && tile != DH_GEMM_TILE_128 && tile != DH_GEMM_TILE_256)
    return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: bad tile request %d", tile);
  if (M < 0 || N < 0 || K < 0) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: negative size");
  if (M == 0 || N == 0) return DH_OK;
  if (!C || (K > 0 && (!A || !B))) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: null operand");
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N)
    return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: leading dimension too small");
  hipStream_t st = dh::as_stream(stream);
  if (tile == DH_GEMM_TILE_AUTO && K > 0 && dh::skinny_applies(M, N, K, trans_a)) {  // narrow layers: HBM-bound streaming kernel (gemm_skinny.hip)
    const int rc = dh::skinny_launch(M, N, K, trans_b, A, lda, B, ldb, C, ldc, accumulate, st);
    if (rc != DH_OK || (!bias && act == DH_ACT_NONE)) return rc;
    return dh_bias_act_f32(M, N, C, ldc, bias, act, stream);
  }
  Plan p = make_plan(M, N, K, tile, trans_a != 0);
  // (Tried in round 5: handing the rows of the last, partial round of large tiles — 134 of 7814 at the headline shape — to the
  // 128 x 128 configuration as a second launch: 14.50 instead of 14.26 ms.  The dispatcher already back-fills the last round as
  // workgroups retire; a second launch adds a drain and a prologue.  profiles/r05l_bench_line.json.)
  float* slabs = nullptr;
  if (p.S > 1) {
    const size_t need = (size_t)p.S * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need)
      return dh::fail(DH_ERR_WORKSPACE, "dh_gemm_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    slabs = static_cast<float*>(workspace);
  }
  // the fast path addresses a block's K-slice with 32-bit lane offsets from the block origin
  const int64_t k_span = (p.k_chunk < K ? p.k_chunk : K) + BK;
  const int64_t span_a = k_span * (trans_a ? lda : 1) * 4 + (int64_t)CfgLarge::BM * lda * 4 + 4096;
  const int64_t span_b = k_span * (trans_b ? 1 : ldb) * 4 + (int64_t)CfgLarge::BN * ldb * 4 + 4096;
  const bool fast_path_ok = span_a < 0xffffffffLL && span_b < 0xffffffffLL;
  const bool aligned = fast_path_ok && dh::aligned16(A) && dh::aligned16(B) && lda % 4 == 0 && ldb % 4 == 0 &&
                       // K-contiguous operands are read 4 k at a time, M/N-contiguous ones are
                       // guarded per element at the edge, so only K % 4 matters for the former
                       ((trans_a != 0 && trans_b == 0) || K % 4 == 0);
  dim3 grid((unsigned)p.n_tiles * (unsigned)p.S);
#define DH_GEMM_LAUNCH(TA, TB, AL)                                                                           \
  do {                                                                                                       \
    if (p.large)                                                                                             \
      hipLaunchKernelGGL((gemm_f32_kernel<CfgLarge, TA, TB, AL>), grid, dim3(CfgLarge::NT), 0, st, M, N, K, A, \
                         lda, B, ldb, C, ldc, accumulate, p.k_chunk, slabs, p.tiles_n, p.n_tiles, p.S, bias, act); \
    else                                                                                                     \
      hipLaunchKernelGGL((gemm_f32_kernel<CfgSmall, TA, TB, AL>), grid, dim3(CfgSmall::NT), 0, st, M, N, K, A, \
                         lda, B, ldb, C, ldc, accumulate, p.k_chunk, slabs, p.tiles_n, p.n_tiles, p.S, bias, act); \
  } while (0)
  const int key = (trans_a ? 4 : 0) | (trans_b ? 2 : 0) | (aligned ? 1 : 0);
  switch (key) {
    case 0: DH_GEMM_LAUNCH(false, false, false); break;
    case 1: DH_GEMM_LAUNCH(false, false, true); break;
    case 2: DH_GEMM_LAUNCH(false, true, false); break;
    case 3: DH_GEMM_LAUNCH(false, true, true); break;
    case 4: DH_GEMM_LAUNCH(true, false, false); break;
    case 5: DH_GEMM_LAUNCH(true, false, true); break;
    case 6: DH_GEMM_LAUNCH(true, true, false); break;
    default: DH_GEMM_LAUNCH(true, true, true); break;
  }
#undef DH_GEMM_LAUNCH
  int rc = dh::check_launch("dh_gemm_f32");
  if (rc != DH_OK) return rc;
  if (p.S > 1) {
    const int64_t total = M * N;
    const unsigned rgrid = (unsigned)(dh::ceil_div(total, 256) < 4096 ? dh::ceil_div(total, 256) : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, st, M, N, p.S, slabs, C, ldc, accumulate, bias, act);
    rc = dh::check_launch("dh_gemm_f32(split-K reduce)");
  }
  return rc;
}
}  // namespace

