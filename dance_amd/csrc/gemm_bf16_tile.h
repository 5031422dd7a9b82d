// 128x128x64 bf16 MFMA tile: C_tile = A[m0:m0+128, k_begin:k_end] · B[n0:n0+128, k_begin:k_end]^T with both operands
// K-contiguous (shared by gemm_bf16.hip and the kNN filter, knn_filter.hip).  4 waves (2x2) of 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16 tiles; LDS images [128][72] bf16 (144-byte rows: the 16-byte fragment reads of 32
// consecutive rows fall on distinct 4-bank slots); the next K-step's global loads are issued before the MFMAs of the
// current one.  `epi(m, n, value, e)` is called once per in-range output element; e = (i * 2 + j) * 16 + r numbers
// the lane's 64 elements (compile-time constant at every call site after unrolling).
#pragma once
#include <type_traits>

#include "common.h"

namespace dh_bf16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDT = BK + 8;  // LDS row stride in bf16 elements (144 bytes)

// round-to-nearest-even f32 -> bf16 bit pattern (NaN stays a quiet NaN)
__device__ __forceinline__ unsigned int f32_to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

constexpr int TILE_LDS_ELEMS = 2 * (BM + BN) * LDT;  // bf16 elements of LDS nt_tile needs: two stages of an A and a B image (73 728 B)

// lds: TILE_LDS_ELEMS bf16 of dynamic shared memory, 16-byte aligned.  Rows >= M / N and k >= k_end read as zero.
// Pipeline: the K-step's operands live in one of two LDS stages; while the MFMAs of step t run, the registers that hold step
// t + 1 (loaded during step t - 1) are written to the other stage and the loads of step t + 2 are issued — one barrier per
// K-step, every memory instruction paired with an MFMA (sched_group_barrier).  Interior tiles (no row / column / K edge)
// load without guards.  (Round 1: one LDS stage, two barriers per step, guarded loads: 693 TFLOP/s at 1M x 2048 x 512.)
template <bool INTERIOR, class Acc>
__device__ __forceinline__ void nt_tile_loop(int64_t M, int64_t N, int64_t k_begin, int64_t k_end, const uint16_t* __restrict__ A,
                                             int64_t lda, const uint16_t* __restrict__ B, int64_t ldb, int64_t m0, int64_t n0,
                                             uint16_t* lds, Acc& acc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // staging: chunk q of this thread covers row (tid / 8 + 32 q), 8 bf16 starting at k = (tid % 8) * 8
  const int srow = tid >> 3, sk = (tid & 7) * 8;
  constexpr int STAGE = (BM + BN) * LDT;
  u32x4 ra[4], rb[4];
  const uint16_t* ap = A + (m0 + srow) * lda + sk;
  const uint16_t* bp = B + (n0 + srow) * ldb + sk;
  auto load_global = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (INTERIOR) {
        ra[q] = *reinterpret_cast<const u32x4*>(ap + (int64_t)32 * q * lda + k0);
        rb[q] = *reinterpret_cast<const u32x4*>(bp + (int64_t)32 * q * ldb + k0);
      } else {
        const int64_t am = m0 + srow + 32 * q, bn = n0 + srow + 32 * q, k = k0 + sk;
        ra[q] = (am < M && k < k_end) ? *reinterpret_cast<const u32x4*>(A + am * lda + k) : u32x4(0u);
        rb[q] = (bn < N && k < k_end) ? *reinterpret_cast<const u32x4*>(B + bn * ldb + k) : u32x4(0u);
      }
    }
  };
  auto store_lds = [&](int stage) __attribute__((always_inline)) {
    uint16_t* As = lds + stage * STAGE;
    uint16_t* Bs = As + BM * LDT;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(As + (srow + 32 * q) * LDT + sk) = ra[q];
      *reinterpret_cast<u32x4*>(Bs + (srow + 32 * q) * LDT + sk) = rb[q];
    }
  };
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const int a_off = (wm * 64 + lr) * LDT + kh, b_off = BM * LDT + (wn * 64 + lr) * LDT + kh;

  const int64_t n_steps = (k_end - k_begin + BK - 1) / BK;
  load_global(k_begin);
  store_lds(0);
  if (n_steps > 1) load_global(k_begin + BK);
  __syncthreads();
  auto step = [&](int64_t t, auto has1_tag, auto has2_tag) __attribute__((always_inline)) {
    constexpr bool HAS1 = decltype(has1_tag)::value, HAS2 = decltype(has2_tag)::value;
    const uint16_t* st = lds + (t & 1) * STAGE;
    // source order = LDS order: this step's fragment reads, then the writes of step t + 1 into the idle stage (last read in
    // step t - 1, before that step's barrier), then the loads of step t + 2 that reuse the staging registers
    bf16x8 a[BK / 16][2], b[BK / 16][2];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[kk][i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 32 * LDT + kk * 16);
        b[kk][i] = *reinterpret_cast<const bf16x8*>(st + b_off + i * 32 * LDT + kk * 16);
      }
    if (HAS1) store_lds((t + 1) & 1);
    if (HAS2) load_global(k_begin + (t + 2) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
    // issue order: 4 fragment reads up front (step kk = 0), then one memory instruction behind every MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (HAS1) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      if (HAS2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
  int64_t t = 0;
  for (; t + 2 < n_steps; ++t) step(t, std::true_type{}, std::true_type{});
  if (t + 1 < n_steps) {
    step(t, std::true_type{}, std::false_type{});
    ++t;
  }
  if (t < n_steps) step(t, std::false_type{}, std::false_type{});
}

template <class Epi>
__device__ __forceinline__ void nt_tile(int64_t M, int64_t N, int64_t k_begin, int64_t k_end, const uint16_t* __restrict__ A,
                                        int64_t lda, const uint16_t* __restrict__ B, int64_t ldb, int64_t m0, int64_t n0,
                                        uint16_t* lds, Epi&& epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (k_end > k_begin) {
    const bool interior = m0 + BM <= M && n0 + BN <= N && (k_end - k_begin) % BK == 0;
    if (interior) nt_tile_loop<true>(M, N, k_begin, k_end, A, lda, B, ldb, m0, n0, lds, acc);
    else nt_tile_loop<false>(M, N, k_begin, k_end, A, lda, B, ldb, m0, n0, lds, acc);
  }

  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + lr;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) epi(m, n, acc[i][j][r], (i * 2 + j) * 16 + r);
      }
    }
}

}  // namespace dh_bf16
