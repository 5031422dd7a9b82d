// Narrow-layer GEMMs (K and N <= 64, one dimension huge): the SpaGCN 50 -> 50 layer, the tail layers of scDSC
// (32 -> n_clusters), cluster heads.  At these widths the 128x128 MFMA tiles of gemm_f32.hip waste > 60 % of the
// matrix pipe and the shapes are HBM-bound anyway (2 flops per byte), so these kernels stream the tall operand
// once through LDS with fully coalesced loads and use plain fp32 FMAs; accumulation order is k (resp. row)
// ascending with one fmaf per term — the same exact-f32 chain semantics as the MFMA kernel.
//
//   rows kernel : C[M,N] (+)= A[M,K] * op(B)   (M huge; torch.mm(x, W) and dX = dS W^T)
// The reduction shape of the same layers (dW = X^T dS, rows huge) stays on the split-K MFMA kernel: a streaming FMA
// form of it measured slower (0.91 vs 0.52 ms at 1M x 50 x 50).
#include "common.h"

namespace dh {

constexpr int SK_MAX = 64;    // max narrow dimension
constexpr int SK_ROWS = 256;  // rows per block iteration

__global__ __launch_bounds__(256) void skinny_rows_kernel(int64_t M, int N, int K, const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb, int trans_b,
                                                          float* __restrict__ C, int64_t ldc, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lda_s = K | 1;               // odd row stride: conflict-free per-row reads
  const int ldn_s = (N + 3) & ~3;        // B rows padded to float4
  const int tile_ld = lda_s > (N | 1) ? lda_s : (N | 1);
  float* As = smem;                      // [256][lda_s]
  float* Bs = As + SK_ROWS * tile_ld;    // [K][ldn_s]   (B[k][n], whatever its storage)
  float* Cs = As;                        // output tile reuses the A tile: [256][N | 1]
  const int tid = threadIdx.x;
  for (int i = tid; i < K * ldn_s; i += 256) {
    const int k = i / ldn_s, n = i % ldn_s;
    Bs[i] = n < N ? (trans_b ? B[(int64_t)n * ldb + k] : B[(int64_t)k * ldb + n]) : 0.f;
  }
  for (int64_t r0 = (int64_t)blockIdx.x * SK_ROWS; r0 < M; r0 += (int64_t)gridDim.x * SK_ROWS) {
    __syncthreads();
    const int rows = (int)min((int64_t)SK_ROWS, M - r0);
    {  // one row per wavefront-load, lane = k: coalesced and free of integer division (K <= 64)
      const int lane = tid & 63;
      if (lane < K)
        for (int r = tid >> 6; r < rows; r += 4) As[r * lda_s + lane] = A[(r0 + r) * lda + lane];
    }
    __syncthreads();
    float acc[SK_MAX];
#pragma unroll
    for (int n = 0; n < SK_MAX; ++n) acc[n] = 0.f;
    if (tid < rows) {
      const float* a = As + tid * lda_s;
      for (int k = 0; k < K; ++k) {
        const float av = a[k];
        const float4* b4 = reinterpret_cast<const float4*>(Bs + k * ldn_s);  // same address in every lane: broadcast
#pragma unroll
        for (int q = 0; q < SK_MAX / 4; ++q) {
          if (q * 4 < N) {
            const float4 b = b4[q];
            acc[q * 4 + 0] = fmaf(av, b.x, acc[q * 4 + 0]);
            acc[q * 4 + 1] = fmaf(av, b.y, acc[q * 4 + 1]);
            acc[q * 4 + 2] = fmaf(av, b.z, acc[q * 4 + 2]);
            acc[q * 4 + 3] = fmaf(av, b.w, acc[q * 4 + 3]);
          }
        }
      }
    }
    __syncthreads();  // everyone is done reading As before it becomes the output tile
    const int ldc_s = N | 1;
    if (tid < rows) {
#pragma unroll
      for (int n = 0; n < SK_MAX; ++n)
        if (n < N) Cs[tid * ldc_s + n] = acc[n];
    }
    __syncthreads();
    {  // coalesced store, one row per wavefront-store
      const int lane = tid & 63;
      if (lane < N)
        for (int r = tid >> 6; r < rows; r += 4) {
          float* p = C + (r0 + r) * ldc + lane;
          *p = accumulate ? *p + Cs[r * ldc_s + lane] : Cs[r * ldc_s + lane];
        }
    }
  }
}

bool skinny_applies(int64_t M, int64_t N, int64_t K, int trans_a) {
  return !trans_a && N <= SK_MAX && K <= SK_MAX && M >= 1024;
}

int skinny_launch(int64_t M, int64_t N, int64_t K, int trans_b, const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, int accumulate, hipStream_t st) {
  static const bool lds_ok = [] {  // tiles may exceed the 64 KiB default dynamic-LDS limit (gfx950 has 160 KiB per CU)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  }();
  if (!lds_ok) return fail(DH_ERR_LAUNCH, "dh_gemm_f32: cannot raise the dynamic LDS limit");
  const int lda_s = (int)K | 1, ldn_s = ((int)N + 3) & ~3;
  const size_t lds = ((size_t)SK_ROWS * (lda_s > ((int)N | 1) ? lda_s : ((int)N | 1)) + (size_t)K * ldn_s) * sizeof(float);
  const unsigned grid = (unsigned)(ceil_div(M, SK_ROWS) < 2048 ? ceil_div(M, SK_ROWS) : 2048);
  hipLaunchKernelGGL(skinny_rows_kernel, dim3(grid), dim3(256), lds, st, M, (int)N, (int)K, A, lda, B, ldb, trans_b, C, ldc, accumulate);
  return check_launch("dh_gemm_f32(narrow rows)");
}

}  // namespace dh
