// 128x128x64 bf16 MFMA tile: C_tile = A[m0:m0+128, k_begin:k_end] · B[n0:n0+128, k_begin:k_end]^T with both operands
// K-contiguous (shared by gemm_bf16.hip and the kNN filter, knn_filter.hip).  4 waves (2x2) of 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16 tiles; LDS images [128][72] bf16 (144-byte rows: the 16-byte fragment reads of 32
// consecutive rows fall on distinct 4-bank slots); the next K-step's global loads are issued before the MFMAs of the
// current one.  `epi(m, n, value, e)` is called once per in-range output element; e = (i * 2 + j) * 16 + r numbers
// the lane's 64 elements (compile-time constant at every call site after unrolling).
#pragma once
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

// As, Bs: __shared__ uint16_t[BM * LDT], [BN * LDT], 16-byte aligned.  Rows >= M / N and k >= k_end read as zero.
template <class Epi>
__device__ __forceinline__ void nt_tile(int64_t M, int64_t N, int64_t k_begin, int64_t k_end, const uint16_t* __restrict__ A,
                                        int64_t lda, const uint16_t* __restrict__ B, int64_t ldb, int64_t m0, int64_t n0,
                                        uint16_t* As, uint16_t* Bs, Epi&& epi) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // staging: chunk q of this thread covers row (tid / 8 + 32 q), 8 bf16 starting at k = (tid % 8) * 8
  const int srow = tid >> 3, sk = (tid & 7) * 8;
  u32x4 ra[4], rb[4];
  auto load_global = [&](int64_t k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t am = m0 + srow + 32 * q, bn = n0 + srow + 32 * q, k = k0 + sk;
      ra[q] = (am < M && k < k_end) ? *reinterpret_cast<const u32x4*>(A + am * lda + k) : u32x4(0u);
      rb[q] = (bn < N && k < k_end) ? *reinterpret_cast<const u32x4*>(B + bn * ldb + k) : u32x4(0u);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<u32x4*>(As + (srow + 32 * q) * LDT + sk) = ra[q];
      *reinterpret_cast<u32x4*>(Bs + (srow + 32 * q) * LDT + sk) = rb[q];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const uint16_t* a_frag = As + (wm * 64 + lr) * LDT + kh;
  const uint16_t* b_frag = Bs + (wn * 64 + lr) * LDT + kh;

  load_global(k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
    store_lds();
    __syncthreads();
    if (k0 + BK < k_end) load_global(k0 + BK);  // in flight behind the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(a_frag + i * 32 * LDT + kk * 16);
        b[i] = *reinterpret_cast<const bf16x8*>(b_frag + i * 32 * LDT + kk * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
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
