// bf16 GEMM on the gfx950 matrix cores — the dense update of the bf16 configuration (SURVEY.md §8a C3:
// "scDeepSort GraphSAGE, bf16 with MFMA dense update"): C = act(op(A) op(B) + bias), bf16 operands, fp32
// accumulation in v_mfma_f32_32x32x16_bf16, output fp32 or bf16 (one round-to-nearest-even).
//
// The kernel proper computes the "NT" form, C[M,N] = A[M,K] · B[N,K]^T, where both operands are K-contiguous: a
// lane's MFMA fragment (8 consecutive k of one row) is then a single 16-byte read, from global memory into the LDS
// image and from the image into registers, with no transposition anywhere.  That is the forward shape of
// torch.nn.Linear (x[M,in] · W[out,in]^T).  An operand stored K-strided (trans_a, or a row-major [K,N] B) is first
// repacked by a tiled bf16 transpose into the caller's workspace (one extra pass over that operand, HBM-bound);
// rows are zero-padded to a multiple of 8 there, which also serves unaligned operands.
//
// 128x128x64 block tile, 4 waves (2x2) of 64x64 = 2x2 MFMA tiles; LDS images [128][72] bf16 (144-B rows: 16-B
// fragment reads of 32 consecutive rows fall on distinct 4-bank slots); the next K-step's global loads are issued
// before the MFMAs of the current one.  Few-tile problems with a long K (dW = dY^T X) are split over K into fp32
// slabs that a second kernel sums in slice order (deterministic, no atomics) and finishes (bias, act, dtype).
#include "gemm_bf16_tile.h"

namespace {

using namespace dh_bf16;

struct Epilogue {
  const float* bias;  // [N] or null
  int act;            // DH_ACT_*
  int accumulate;     // C += result (read in C's dtype)
  int c_bf16;         // output dtype
};

__device__ __forceinline__ void store_out(void* C, int64_t ldc, int64_t m, int64_t n, float v, const Epilogue& ep) {
  if (ep.bias) v += ep.bias[n];
  if (ep.act == DH_ACT_RELU) v = fmaxf(v, 0.f);
  if (ep.c_bf16) {
    uint16_t* p = static_cast<uint16_t*>(C) + m * ldc + n;
    if (ep.accumulate) v += __uint_as_float((unsigned int)*p << 16);
    *p = (uint16_t)f32_to_bf16(v);
  } else {
    float* p = static_cast<float*>(C) + m * ldc + n;
    if (ep.accumulate) v += *p;
    *p = v;
  }
}

// grid: x = tiles (n fastest), z = K slices.  slabs != null: write the raw fp32 partial of this slice.
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(int64_t M, int64_t N, int64_t K, const uint16_t* __restrict__ A,
                                                           int64_t lda, const uint16_t* __restrict__ B, int64_t ldb,
                                                           void* __restrict__ C, int64_t ldc, Epilogue ep,
                                                           float* __restrict__ slabs, int64_t k_per_slice) {
  extern __shared__ __attribute__((aligned(16))) uint16_t tile_lds[];  // TILE_LDS_ELEMS bf16 (two pipeline stages)
  const int64_t tiles_n = (N + BN - 1) / BN;
  // XCD-aware bijective remap (the dispatcher places block b on XCD b % 8, observed): every XCD gets a contiguous run of
  // logical tiles, so the column tiles that share an A row panel share one L2 (with the plain order each of the 8 XCDs
  // fetched every panel itself: 1M x 2048 x 512 ran HBM-bound at 690 TFLOP/s)
  const int64_t n_tiles = gridDim.x, bid = blockIdx.x;
  const int64_t q = n_tiles / 8, rr = n_tiles % 8, xcd = bid % 8;
  const int64_t logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + bid / 8;
  // inside an XCD's run: groups of 8 tile rows x all tile columns, rows fastest, so the ~64 tiles an XCD runs at a time form a
  // patch that shares 8 A panels and a few B panels instead of one A panel and every B panel
  const int64_t tiles_m = n_tiles / tiles_n, GM = 8;
  const int64_t group = logical / (GM * tiles_n), first_m = group * GM, in_group = logical % (GM * tiles_n);
  const int64_t gm = min(GM, tiles_m - first_m);
  const int64_t m0 = (first_m + in_group % gm) * BM, n0 = (in_group / gm) * BN;
  const int64_t k_begin = (int64_t)blockIdx.z * k_per_slice;
  const int64_t k_end = min(K, k_begin + k_per_slice);
  nt_tile(M, N, k_begin, k_end, A, lda, B, ldb, m0, n0, tile_lds, [&](int64_t m, int64_t n, float v, int) {
    if (slabs) slabs[((int64_t)blockIdx.z * M + m) * N + n] = v;
    else store_out(C, ldc, m, n, v, ep);
  });
}

__global__ __launch_bounds__(256) void gemm_bf16_reduce_kernel(int64_t M, int64_t N, int slices, const float* __restrict__ slabs,
                                                               void* __restrict__ C, int64_t ldc, Epilogue ep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  float v = 0.f;
  for (int z = 0; z < slices; ++z) v += slabs[(int64_t)z * M * N + i];
  store_out(C, ldc, i / N, i % N, v, ep);
}

// out[c][r] = in[r][c] for r < rows, c < cols; out rows are ld_out long and zero beyond `rows` (pad to 8).
// With transpose == 0 it is a padding copy: out[r][c] = in[r][c], zero beyond `cols`.
__global__ __launch_bounds__(256) void repack_bf16_kernel(int64_t rows, int64_t cols, const uint16_t* __restrict__ in, int64_t ld_in,
                                                          uint16_t* __restrict__ out, int64_t ld_out, int transpose) {
  __shared__ uint16_t tile[64][66];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  if (!transpose) {
    for (int y = ty; y < 64; y += 4) {
      const int64_t r = r0 + y, c = c0 + tx;
      if (r < rows && c < ld_out) out[r * ld_out + c] = c < cols ? in[r * ld_in + c] : (uint16_t)0;
    }
    return;
  }
  for (int y = ty; y < 64; y += 4) {
    const int64_t r = r0 + y, c = c0 + tx;
    tile[y][tx] = (r < rows && c < cols) ? in[r * ld_in + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int y = ty; y < 64; y += 4) {
    const int64_t c = c0 + y, r = r0 + tx;  // out row = input column
    if (c < cols && r < ld_out) out[c * ld_out + r] = tile[tx][y];  // r in [rows, ld_out) carries the zero fill
  }
}

struct Plan {
  bool repack_a, repack_b;
  int64_t kp;           // padded K (leading dimension of repacked operands)
  int slices;
  int64_t k_per_slice;
  size_t a_bytes, b_bytes, slab_bytes;
};

size_t round256(size_t b) { return (b + 255) / 256 * 256; }

Plan make_plan(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb) {
  Plan p{};
  p.kp = (K + 7) / 8 * 8;
  // native = K-contiguous, 16-byte aligned rows, K a multiple of 8 (A == null: the size query, which assumes that
  // K-contiguous operands ARE aligned — the caller aligns them or gets DH_ERR_WORKSPACE)
  p.repack_a = trans_a || K % 8 != 0 || (A && (lda % 8 != 0 || !dh::aligned16(A)));
  p.repack_b = !trans_b || K % 8 != 0 || (B && (ldb % 8 != 0 || !dh::aligned16(B)));
  p.a_bytes = p.repack_a ? round256((size_t)M * p.kp * 2) : 0;
  p.b_bytes = p.repack_b ? round256((size_t)N * p.kp * 2) : 0;
  const int64_t tiles = dh::ceil_div(M, BM) * dh::ceil_div(N, BN);
  int64_t slices = 1;
  if (tiles < 512 && K >= 8 * BK) {
    slices = 1024 / tiles;
    const int64_t max_slices = K / (4 * BK);  // at least 4 K-steps per slice
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
  }
  p.k_per_slice = dh::ceil_div(dh::ceil_div(K, slices), BK) * BK;
  p.slices = (int)dh::ceil_div(K, p.k_per_slice);
  p.slab_bytes = p.slices > 1 ? round256((size_t)p.slices * M * N * sizeof(float)) : 0;
  return p;
}

}  // namespace

extern "C" size_t dh_gemm_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan p = make_plan(M, N, K, trans_a, trans_b, nullptr, 0, nullptr, 0);
  return p.a_bytes + p.b_bytes + p.slab_bytes;
}

extern "C" int dh_gemm_bf16(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const uint16_t* A, int64_t lda,
                            const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int c_dtype, const float* bias, int act,
                            int accumulate, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return dh::fail(DH_ERR_INVALID, "dh_gemm_bf16: negative size");
  if (M == 0 || N == 0) return DH_OK;
  if (!C || (K > 0 && (!A || !B))) return dh::fail(DH_ERR_INVALID, "dh_gemm_bf16: null pointer");
  if (c_dtype != DH_DTYPE_F32 && c_dtype != DH_DTYPE_BF16) return dh::fail(DH_ERR_INVALID, "dh_gemm_bf16: bad output dtype %d", c_dtype);
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_gemm_bf16: bad act %d", act);
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) return dh::fail(DH_ERR_INVALID, "dh_gemm_bf16: leading dimension too small");
  hipStream_t st = dh::as_stream(stream);
  const Epilogue ep{bias, act, accumulate, c_dtype == DH_DTYPE_BF16};
  if (K == 0) {  // empty sum: C = act(bias) (+ C)
    hipLaunchKernelGGL(gemm_bf16_reduce_kernel, dim3((unsigned)dh::ceil_div(M * N, 256)), dim3(256), 0, st, M, N, 0, nullptr, C, ldc, ep);
    return dh::check_launch("dh_gemm_bf16");
  }
  const Plan p = make_plan(M, N, K, trans_a, trans_b, A, lda, B, ldb);
  const size_t need = p.a_bytes + p.b_bytes + p.slab_bytes;
  if (need && (!workspace || workspace_bytes < need || !dh::aligned16(workspace)))
    return dh::fail(DH_ERR_WORKSPACE, "dh_gemm_bf16: workspace %zu < %zu bytes", workspace_bytes, need);
  char* ws = static_cast<char*>(workspace);
  const uint16_t* a = A;
  const uint16_t* b = B;
  int64_t la = lda, lb = ldb;
  if (p.repack_a) {
    uint16_t* ap = reinterpret_cast<uint16_t*>(ws);
    // trans_a: A is stored [K][M] -> [M][kp]; else a padding copy of [M][K]
    const int64_t rows = trans_a ? K : M, cols = trans_a ? M : K;
    dim3 grid((unsigned)dh::ceil_div(trans_a ? cols : p.kp, 64), (unsigned)dh::ceil_div(trans_a ? p.kp : rows, 64));
    hipLaunchKernelGGL(repack_bf16_kernel, grid, dim3(256), 0, st, rows, cols, A, lda, ap, p.kp, trans_a ? 1 : 0);
    a = ap; la = p.kp;
  }
  if (p.repack_b) {
    uint16_t* bp = reinterpret_cast<uint16_t*>(ws + p.a_bytes);
    // !trans_b: B is stored [K][N] -> [N][kp]; else a padding copy of [N][K]
    const int64_t rows = trans_b ? N : K, cols = trans_b ? K : N;
    dim3 grid((unsigned)dh::ceil_div(trans_b ? p.kp : cols, 64), (unsigned)dh::ceil_div(trans_b ? rows : p.kp, 64));
    hipLaunchKernelGGL(repack_bf16_kernel, grid, dim3(256), 0, st, rows, cols, B, ldb, bp, p.kp, trans_b ? 0 : 1);
    b = bp; lb = p.kp;
  }
  float* slabs = p.slices > 1 ? reinterpret_cast<float*>(ws + p.a_bytes + p.b_bytes) : nullptr;
  dim3 grid((unsigned)(dh::ceil_div(M, BM) * dh::ceil_div(N, BN)), 1, (unsigned)p.slices);
  constexpr size_t kTileLds = (size_t)TILE_LDS_ELEMS * sizeof(uint16_t);  // 72 KB: above the 64 KB default limit of dynamic LDS
  static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_nt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)kTileLds) == hipSuccess;
  if (!lds_ok) return dh::fail(DH_ERR_LAUNCH, "dh_gemm_bf16: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(gemm_bf16_nt_kernel, grid, dim3(256), kTileLds, st, M, N, K, a, la, b, lb, C, ldc, ep, slabs, p.k_per_slice);
  if (slabs)
    hipLaunchKernelGGL(gemm_bf16_reduce_kernel, dim3((unsigned)dh::ceil_div(M * N, 256)), dim3(256), 0, st, M, N, p.slices, slabs, C, ldc, ep);
  return dh::check_launch("dh_gemm_bf16");
}
