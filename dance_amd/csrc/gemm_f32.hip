// Dense feature GEMM on the CDNA4 f32 matrix cores.
//
// C[M,N] (+)= op(A)[M,K] * op(B)[K,N], exact f32: v_mfma_f32_32x32x2_f32 is bit-for-bit a
// k-ordered fmaf chain (one rounding per product), so the layer keeps the reference's fp32
// semantics (torch.mm, scdsc.py:497 / spagcn.py:358) — gfx950 has no TF32-like mode and we
// do not down-convert.  MFMA-bound: 157 TFLOP/s peak for f32 inputs.
//
// Block = 4 wavefronts computing a 128x128 tile; each wavefront owns 64x64 = 2x2 MFMA tiles
// (64 accumulator VGPRs).  K is consumed 32 at a time through a register-staged,
// double-buffered LDS pipeline (global loads of tile t+1 are in flight while tile t feeds
// the matrix cores; one barrier per K-step).  LDS images are padded so the fragment reads
// are bank-conflict free:
//   "MK" image (operand stored with K contiguous): [128][32+4] floats, fragment = one
//        ds_read_b128 per lane (4 consecutive k of one row) -> 4 MFMA steps;
//   "KM" image (operand stored with M/N contiguous): [32][128+4] floats, fragment = 4
//        ds_read_b32 (rows k, k+1.. of one column).
// Lane half h = lane>>5 supplies k = 8*g + 4*h + s at MFMA step s of k-group g for BOTH
// operands, so the hardware's (k = lane>>5) pairing is a permutation of the K-slice.
//
// The transposed-A form (dW = X^T dZ, K = number of cells) is split over K across
// gridDim.z; partial slabs are summed by a second deterministic kernel (no float atomics).
// Block ids are remapped so that consecutive tiles land on the same XCD (private L2) and
// share their A row panel.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_MK = BK + 4;    // 36 floats: 16-B aligned rows, b128 reads conflict-free
constexpr int LD_KM = BM + 4;    // 132 floats
constexpr int TILE_MK = BM * LD_MK;  // 4608 floats
constexpr int TILE_KM = BK * LD_KM;  // 4224 floats
constexpr int TILE_MAX = TILE_MK > TILE_KM ? TILE_MK : TILE_KM;

// Global -> registers for one operand tile.  KCONTIG: operand stored [rows][K] (k contiguous);
// otherwise stored [K][rows].  `rows` is the M (or N) extent, r0 the tile's first row.
template <bool KCONTIG, bool ALIGNED>
__device__ __forceinline__ void load_tile(f32x4 (&st)[4], const float* __restrict__ P, int64_t ld,
                                          int64_t rows, int64_t r0, int64_t k0, int64_t k_end,
                                          int tid) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = tid + 256 * r;
    f32x4 v = f32x4(0.f);
    if constexpr (KCONTIG) {
      const int row = idx >> 3, kq = (idx & 7) * 4;
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
      const int kk = idx >> 5, mq = (idx & 31) * 4;
      const int64_t gk = k0 + kk;
      const int64_t gm = r0 + mq;
      if (gk < k_end) {
        const float* p = P + gk * ld + gm;
        if constexpr (ALIGNED) {
          if (gm + 3 < rows) v = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (gm + i < rows) v[i] = p[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (gm + i < rows) v[i] = p[i];
        }
      }
    }
    st[r] = v;
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const f32x4 (&st)[4], int tid) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = tid + 256 * r;
    if constexpr (KCONTIG) {
      const int row = idx >> 3, kq = (idx & 7) * 4;
      *reinterpret_cast<f32x4*>(lds + row * LD_MK + kq) = st[r];
    } else {
      const int kk = idx >> 5, mq = (idx & 31) * 4;
      *reinterpret_cast<f32x4*>(lds + kk * LD_KM + mq) = st[r];
    }
  }
}

// Fragment for MFMA steps s = 0..3 of k-group g: element s = operand[row][8g + 4h + s].
template <bool KCONTIG>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ lds, int row, int g, int h) {
  if constexpr (KCONTIG) {
    return *reinterpret_cast<const f32x4*>(lds + row * LD_MK + g * 8 + h * 4);
  } else {
    f32x4 v;
    const float* p = lds + (g * 8 + h * 4) * LD_KM + row;
    v[0] = p[0]; v[1] = p[LD_KM]; v[2] = p[2 * LD_KM]; v[3] = p[3 * LD_KM];
    return v;
  }
}

// TA: A stored [K][M]; TB: B stored [N][K].
template <bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(
    int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int accumulate, int64_t k_chunk, float* __restrict__ slabs, int tiles_n, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_MAX];
  // buffers: A[0], A[1], B[0], B[1]

  // XCD-aware bijective remap: the dispatcher places block b on XCD b % 8; give every XCD a
  // contiguous run of logical tiles so neighbours (same A row panel) share one L2.
  const int bid = blockIdx.x;
  const int q = n_tiles / 8, rr = n_tiles % 8, xcd = bid % 8;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + bid / 8;
  const int64_t m0 = (int64_t)(logical / tiles_n) * BM;
  const int64_t n0 = (int64_t)(logical % tiles_n) * BN;

  const int64_t k_begin = (int64_t)blockIdx.z * k_chunk;
  const int64_t k_end = min(K, k_begin + k_chunk);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i32 = lane & 31, h = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x16(0.f);

  f32x4 sa[4], sb[4];
  const int64_t n_steps = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;
  if (n_steps > 0) {
    load_tile<!TA, ALIGNED>(sa, A, lda, M, m0, k_begin, k_end, tid);
    load_tile<TB, ALIGNED>(sb, B, ldb, N, n0, k_begin, k_end, tid);
    store_tile<!TA>(lds, sa, tid);
    store_tile<TB>(lds + 2 * TILE_MAX, sb, tid);
  }
  __syncthreads();

  for (int64_t t = 0; t < n_steps; ++t) {
    const int cur = t & 1;
    if (t + 1 < n_steps) {
      const int64_t k0 = k_begin + (t + 1) * BK;
      load_tile<!TA, ALIGNED>(sa, A, lda, M, m0, k0, k_end, tid);
      load_tile<TB, ALIGNED>(sb, B, ldb, N, n0, k0, k_end, tid);
    }
    const float* a_lds = lds + cur * TILE_MAX;
    const float* b_lds = lds + (2 + cur) * TILE_MAX;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        fa[x] = read_frag<!TA>(a_lds, wm * 64 + x * 32 + i32, g, h);
        fb[x] = read_frag<TB>(b_lds, wn * 64 + x * 32 + i32, g, h);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x][s], fb[y][s], acc[x][y], 0, 0, 0);
    }
    if (t + 1 < n_steps) {
      store_tile<!TA>(lds + (cur ^ 1) * TILE_MAX, sa, tid);
      store_tile<TB>(lds + (2 + (cur ^ 1)) * TILE_MAX, sb, tid);
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
  float* out = C;
  int64_t ldo = ldc;
  bool add = accumulate != 0;
  if (slabs) {
    out = slabs + (int64_t)blockIdx.z * M * N;
    ldo = N;
    add = false;
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int64_t col = n0 + wn * 64 + y * 32 + i32;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= M) continue;
        float* p = out + row * ldo + col;
        *p = add ? (*p + acc[x][y][r]) : acc[x][y][r];
      }
    }
}

// C = (accumulate ? C : 0) + sum_z slabs[z]   (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t M, int64_t N, int S,
                                                            const float* __restrict__ slabs,
                                                            float* __restrict__ C, int64_t ldc,
                                                            int accumulate) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / N, col = i % N;
    float* p = C + row * ldc + col;
    float s = accumulate ? *p : 0.f;
    for (int z = 0; z < S; ++z) s += slabs[(int64_t)z * total + i];
    *p = s;
  }
}

struct Plan {
  int tiles_m, tiles_n, n_tiles, S;
  int64_t k_chunk;
};

Plan make_plan(int64_t M, int64_t N, int64_t K) {
  Plan p;
  p.tiles_m = (int)dh::ceil_div(M, BM);
  p.tiles_n = (int)dh::ceil_div(N, BN);
  p.n_tiles = p.tiles_m * p.tiles_n;
  p.S = 1;
  p.k_chunk = dh::ceil_div(K > 0 ? K : 1, BK) * BK;
  // Few output tiles and a long K (dW = X^T dZ): split K until ~4 blocks per CU are in flight,
  // keeping at least 64 K-steps per block.
  if (p.n_tiles < 512 && K >= 4096) {
    int64_t want = dh::ceil_div(1024, p.n_tiles);
    int64_t max_s = K / (64 * BK);
    if (max_s < 1) max_s = 1;
    int64_t S = want < max_s ? want : max_s;
    if (S > 1) {
      p.k_chunk = dh::ceil_div(dh::ceil_div(K, S), BK) * BK;
      p.S = (int)dh::ceil_div(K, p.k_chunk);
    }
  }
  return p;
}

}  // namespace

extern "C" size_t dh_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b) {
  (void)trans_a; (void)trans_b;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  Plan p = make_plan(M, N, K);
  return p.S > 1 ? (size_t)p.S * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int dh_gemm_f32(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                           int64_t ldc, int accumulate, void* workspace, size_t workspace_bytes,
                           dh_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: negative size");
  if (M == 0 || N == 0) return DH_OK;
  if (!C || (K > 0 && (!A || !B))) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: null operand");
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N)
    return dh::fail(DH_ERR_INVALID, "dh_gemm_f32: leading dimension too small");
  hipStream_t st = dh::as_stream(stream);
  Plan p = make_plan(M, N, K);
  float* slabs = nullptr;
  if (p.S > 1) {
    const size_t need = (size_t)p.S * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need)
      return dh::fail(DH_ERR_WORKSPACE, "dh_gemm_f32: workspace %zu < %zu bytes", workspace_bytes, need);
    slabs = static_cast<float*>(workspace);
  }
  const bool aligned = dh::aligned16(A) && dh::aligned16(B) && lda % 4 == 0 && ldb % 4 == 0 &&
                       // K-contiguous operands are read 4 k at a time, M/N-contiguous ones are
                       // guarded per element at the edge, so only K % 4 matters for the former
                       ((trans_a != 0 && trans_b == 0) || K % 4 == 0);
  dim3 grid((unsigned)p.n_tiles, 1, (unsigned)p.S), block(256);
#define DH_GEMM_LAUNCH(TA, TB, AL)                                                              \
  hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, AL>), grid, block, 0, st, M, N, K, A, lda, B, ldb, \
                     C, ldc, accumulate, p.k_chunk, slabs, p.tiles_n, p.n_tiles)
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
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, st, M, N, p.S, slabs, C, ldc, accumulate);
    rc = dh::check_launch("dh_gemm_f32(split-K reduce)");
  }
  return rc;
}
