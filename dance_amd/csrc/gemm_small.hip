// Small fp32 GEMMs of the mini-batch steps: C[M,N] = act(op(A) op(B) + bias) for K <= 512 and M N <= 2^20, exact fp32 on the matrix
// cores (v_mfma_f32_32x32x2_f32, the arithmetic of gemm_f32.hip in another summation order).
//
// Why a second kernel: a graph-sc step at the reference's batch size (128 cells + 2000 genes) runs seven products of this size —
// [2128 x 50][50 x 200] twice, [128 x 200][300 x 200]^T twice, [128 x 300][300 x 200], [128 x 300]^T[128 x 200], and one long-K dW — and
// through the 128 x 128 tiles of gemm_f32.hip each is 1 - 34 workgroups walking K in 32-wide steps, one dependent global round trip per
// step with nothing else resident to hide it: 18 - 32 us apiece, 0.17 of the 0.39 ms step (profiles/r05u_graphsc_step_kernels_b128.md),
// as much as its 59 other launches together.  Here a workgroup owns a 32 x 32 tile, fetches its WHOLE K extent (up to 256 per pass) with
// every load in flight at once, and its four wavefronts take interleaved k-steps; their partial tiles meet in LDS in wave order
// (deterministic).  One round trip, ~K / 8 matrix instructions per wave, a coalesced store with nn.Linear's bias / ReLU folded in.
//
// LDS: both panels k-major, As[k][i] / Bs[k][j] with a row stride of 33 floats: the MFMA operand read (lane = (i, k half)) walks
// consecutive addresses, and the staging writes — consecutive lanes carry consecutive k for a K-contiguous operand, consecutive i / j
// for the other kind — land on distinct banks either way.  No load sits behind a divergent branch (clamped address + select).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GS_T = 32;   // tile edge
constexpr int GS_LD = 33;  // LDS row stride

// panel[k * 33 + r] = op(P)(row0 + r, k0 + k) for k < KC, r < 32; zero outside the matrix.  KCONTIG: memory runs along k
// (P[(row0 + r) * ld + k0 + k]: A of NN / NT, B of NT), else along r (P[(k0 + k) * ld + row0 + r]: A of TN, B of NN).
template <int KC, bool KCONTIG>
__device__ __forceinline__ void gs_load(float (&reg)[KC / 8], const float* __restrict__ P, int64_t ld, int row0, int n_rows, int k0, int K) {
#pragma unroll
  for (int q = 0; q < KC / 8; ++q) {
    const int e = threadIdx.x + 256 * q;
    const int r = KCONTIG ? e / KC : e % GS_T, k = KCONTIG ? e % KC : e / GS_T;
    const int rr = min(row0 + r, n_rows - 1), kk = min(k0 + k, K - 1);
    reg[q] = KCONTIG ? P[(int64_t)rr * ld + kk] : P[(int64_t)kk * ld + rr];
  }
}
template <int KC, bool KCONTIG>
__device__ __forceinline__ void gs_store(const float (&reg)[KC / 8], float* panel, int row0, int n_rows, int k0, int K) {
#pragma unroll
  for (int q = 0; q < KC / 8; ++q) {
    const int e = threadIdx.x + 256 * q;
    const int r = KCONTIG ? e / KC : e % GS_T, k = KCONTIG ? e % KC : e / GS_T;
    panel[k * GS_LD + r] = (row0 + r < n_rows && k0 + k < K) ? reg[q] : 0.f;
  }
}

template <int KC, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_small_kernel(int M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                         int64_t ldb, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias, int act) {
  __shared__ float smem[2 * KC * GS_LD];  // A panel, B panel; afterwards the four partial tiles (4 x 32 x 33 floats <= 2 x 64 x 33)
  float* const As = smem;
  float* const Bs = smem + KC * GS_LD;
  const int m0 = blockIdx.y * GS_T, n0 = blockIdx.x * GS_T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += KC) {
    float ra[KC / 8], rb[KC / 8];
    gs_load<KC, !TA>(ra, A, lda, m0, M, k0, K);   // op(A)(i, k): A[i][k] (K-contiguous) or A[k][i]
    gs_load<KC, TB>(rb, B, ldb, n0, N, k0, K);    // op(B)(k, j): B[k][j] or B[j][k] (K-contiguous)
    if (k0) __syncthreads();                      // the previous pass's fragments have been read
    gs_store<KC, !TA>(ra, As, m0, M, k0, K);
    gs_store<KC, TB>(rb, Bs, n0, N, k0, K);
    __syncthreads();
    const int kc = min(KC, K - k0);
    for (int s = wave; 2 * s < kc; s += 4) {      // k-steps of 2, dealt round-robin to the four waves
      const float a = As[(2 * s + h) * GS_LD + i32];
      const float b = Bs[(2 * s + h) * GS_LD + i32];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  // C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* const red = smem + wave * (GS_T * GS_LD);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * h) * GS_LD + i32] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = threadIdx.x + 256 * q;
    const int row = e >> 5, col = e & 31;
    float v = smem[row * GS_LD + col];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += smem[w * (GS_T * GS_LD) + row * GS_LD + col];  // wave order: deterministic
    if (m0 + row < M && n0 + col < N) {
      if (bias) v += bias[n0 + col];
      if (act == DH_ACT_RELU) v = fmaxf(v, 0.f);
      C[(int64_t)(m0 + row) * ldc + n0 + col] = v;
    }
  }
}

template <int KC>
void gs_launch(int ta, int tb, dim3 grid, hipStream_t st, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
               int64_t ldc, const float* bias, int act) {
  if (!ta && !tb) hipLaunchKernelGGL((gemm_small_kernel<KC, false, false>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, act);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_small_kernel<KC, false, true>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, act);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_small_kernel<KC, true, false>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, act);
  else hipLaunchKernelGGL((gemm_small_kernel<KC, true, true>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, act);
}

}  // namespace

extern "C" int dh_gemm_f32_small_supported(int64_t M, int64_t N, int64_t K) {
  return (M >= 1 && N >= 1 && K >= 1 && K <= 512 && M * N <= ((int64_t)1 << 20)) ? 1 : 0;
}

extern "C" int dh_gemm_f32_small(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B,
                                 int64_t ldb, float* C, int64_t ldc, const float* bias, int act, dh_stream_t stream) {
  const char* me = "dh_gemm_f32_small";
  if (M < 0 || N < 0 || K < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (M == 0 || N == 0) return DH_OK;
  if (!dh_gemm_f32_small_supported(M, N, K))
    return dh::fail(DH_ERR_INVALID, "%s: %lld x %lld x %lld outside K in [1, 512], M N <= 2^20 (dh_gemm_f32_small_supported; use dh_gemm_f32)", me,
                    (long long)M, (long long)N, (long long)K);
  if (!A || !B || !C) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) return dh::fail(DH_ERR_INVALID, "%s: leading dimension too small", me);
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "%s: bad act %d", me, act);
  hipStream_t st = dh::as_stream(stream);
  const dim3 grid((unsigned)dh::ceil_div(N, GS_T), (unsigned)dh::ceil_div(M, GS_T));
  if (K <= 64) gs_launch<64>(trans_a, trans_b, grid, st, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, bias, act);
  else if (K <= 128) gs_launch<128>(trans_a, trans_b, grid, st, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, bias, act);
  else gs_launch<256>(trans_a, trans_b, grid, st, (int)M, (int)N, (int)K, A, lda, B, ldb, C, ldc, bias, act);
  return dh::check_launch(me);
}
