// Narrow-layer GEMMs (K and N <= 64, one dimension huge): the SpaGCN 50 -> 50 layer, the tail layers of scDSC
// (32 -> n_clusters), cluster heads.  At these widths the 128x128 MFMA tiles of gemm_f32.hip waste > 60 % of the
// matrix pipe and the shapes are HBM-bound anyway (2 flops per byte), so these kernels stream the tall operand
// once through LDS with fully coalesced loads and use plain fp32 FMAs; accumulation order is k (resp. row)
// ascending with one fmaf per term — the same exact-f32 chain semantics as the MFMA kernel.
//
//   rows kernel   : C[M,N] (+)= A[M,K] * op(B)      (M huge; torch.mm(x, W) and dX = dS W^T)
//   reduce kernel : C[Kd,N] (+)= A[R,Kd]^T * B[R,N]  (R huge; dW = X^T dS) — per-block partial slabs + the
//                   deterministic slab reduction shared with the split-K MFMA path.
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

// partial[b][m][n] = sum over this block's rows of A[r][m] * B[r][n];  thread = (m, 16-wide n group)
__global__ __launch_bounds__(256) void skinny_reduce_kernel(int64_t R, int Md, int N, const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ B, int64_t ldb, int64_t rows_per_block,
                                                            float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lda_s = Md | 1;
  const int ldb_s = ((N + 15) & ~15) + 4;  // 16-float groups, float4 aligned
  float* As = smem;                        // [256][lda_s]
  float* Bs = As + SK_ROWS * lda_s;        // [256][ldb_s]
  const int tid = threadIdx.x, m = tid >> 2, ng = tid & 3;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int64_t rb = (int64_t)blockIdx.x * rows_per_block, re = min(R, rb + rows_per_block);
  for (int64_t r0 = rb; r0 < re; r0 += SK_ROWS) {
    __syncthreads();
    const int rows = (int)min((int64_t)SK_ROWS, re - r0);
    for (int i = tid; i < rows * Md; i += 256) {
      const int r = i / Md, c = i % Md;
      As[r * lda_s + c] = A[(r0 + r) * lda + c];
    }
    for (int i = tid; i < rows * (ldb_s - 4); i += 256) {
      const int r = i / (ldb_s - 4), c = i % (ldb_s - 4);
      Bs[r * ldb_s + c] = c < N ? B[(r0 + r) * ldb + c] : 0.f;
    }
    __syncthreads();
    if (m < Md && ng * 16 < N) {
      for (int r = 0; r < rows; ++r) {
        const float av = As[r * lda_s + m];
        const float4* b4 = reinterpret_cast<const float4*>(Bs + r * ldb_s + ng * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = b4[q];
          acc[q * 4 + 0] = fmaf(av, b.x, acc[q * 4 + 0]);
          acc[q * 4 + 1] = fmaf(av, b.y, acc[q * 4 + 1]);
          acc[q * 4 + 2] = fmaf(av, b.z, acc[q * 4 + 2]);
          acc[q * 4 + 3] = fmaf(av, b.w, acc[q * 4 + 3]);
        }
      }
    }
  }
  if (m < Md) {
    float* out = slabs + ((int64_t)blockIdx.x * Md + m) * N;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (ng * 16 + i < N) out[ng * 16 + i] = acc[i];
  }
}

constexpr int64_t kSkinnyRowsPerBlock = 4096;

bool skinny_applies(int64_t M, int64_t N, int64_t K, int trans_a) {
  // The reduction form (dW of a narrow layer) measured slower than the split-K MFMA kernel (0.91 vs 0.52 ms at
  // 1M x 50 x 50), so it is kept for reference but not dispatched.
  if (trans_a) return false;
  return N <= SK_MAX && K <= SK_MAX && M >= 1024;
}

size_t skinny_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a) {
  if (!trans_a) return 0;
  return (size_t)ceil_div(K, kSkinnyRowsPerBlock) * (size_t)M * (size_t)N * sizeof(float);
}

int skinny_slab_count(int64_t K) { return (int)ceil_div(K, kSkinnyRowsPerBlock); }

// launches the narrow kernels; for the reduction form the caller runs the slab reduction afterwards
int skinny_launch(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B,
                  int64_t ldb, float* C, int64_t ldc, int accumulate, float* slabs, hipStream_t st) {
  static const bool lds_ok = [] {  // tiles may exceed the 64 KiB default dynamic-LDS limit (gfx950 has 160 KiB per CU)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  }();
  if (!lds_ok) return fail(DH_ERR_LAUNCH, "dh_gemm_f32: cannot raise the dynamic LDS limit");
  if (!trans_a) {
    const int lda_s = (int)K | 1, ldn_s = ((int)N + 3) & ~3;
    const size_t lds = ((size_t)SK_ROWS * (lda_s > ((int)N | 1) ? lda_s : ((int)N | 1)) + (size_t)K * ldn_s) * sizeof(float);
    const unsigned grid = (unsigned)(ceil_div(M, SK_ROWS) < 2048 ? ceil_div(M, SK_ROWS) : 2048);
    hipLaunchKernelGGL(skinny_rows_kernel, dim3(grid), dim3(256), lds, st, M, (int)N, (int)K, A, lda, B, ldb, trans_b, C, ldc, accumulate);
    return check_launch("dh_gemm_f32(narrow rows)");
  }
  if (trans_b) return fail(DH_ERR_INVALID, "dh_gemm_f32: narrow A^T B^T is not a layer shape");
  const int lda_s = (int)M | 1, ldb_s = (((int)N + 15) & ~15) + 4;
  const size_t lds = (size_t)SK_ROWS * (lda_s + ldb_s) * sizeof(float);
  hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)skinny_slab_count(K)), dim3(256), lds, st, K, (int)M, (int)N, A, lda, B,
                     ldb, kSkinnyRowsPerBlock, slabs);
  return check_launch("dh_gemm_f32(narrow reduce)");
}

}  // namespace dh
