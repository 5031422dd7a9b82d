// fp32 GEMM on the bf16 matrix cores by operand splitting ("bf16x3"): the dense feature GEMM of the GCN layer
// (scdsc.py:497 `torch.mm(features, self.weight)`, spagcn.py:358) and its weight gradient dW = X^T dZ.
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s while v_mfma_f32_32x32x16_bf16 peaks at 2.5 PFLOP/s — 16x.
// Every fp32 value is the exact sum of three bf16 values,  x = x_h + x_m + x_l  (x_h = bf16(x), x_m = bf16(x - x_h),
// x_l = bf16(x - x_h - x_m); the residuals are exact in fp32 and 8 + 8 + 8 significant bits cover the 24-bit mantissa), so
//   x * w = x_h w_h + x_h w_m + x_m w_h + x_m w_m + x_h w_l + x_l w_h  +  (x_m w_l + x_l w_m + x_l w_l)
// and the three dropped products are <= 2^-23 |x w| together — below the rounding of ONE fp32 addition, i.e. far inside the
// error a K = 2000 fp32 accumulation already has.  Each bf16 x bf16 product is exact in fp32 and the matrix core accumulates
// in fp32, so the result has fp32 accuracy (tests/test_gpu_gemm_x3.py measures it against float64 next to the exact-fp32
// kernel) at 6/16 of the matrix-pipe time.  Inputs must be finite (inf - inf in the residual turns an inf into NaN).
//
// Kernel: 256 x 256 block tile, 8 wavefronts (2 x 4), each 128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers);
// K is consumed 16 at a time.  The fp32 operand tiles are loaded straight from global memory into registers (buffer loads,
// scalar descriptor + loop-invariant lane offset), split into the three bf16 planes in registers, and written to LDS
// TRANSPOSED INTO FRAGMENT ORDER:  plane[p][k-group (8 k)][row][8 bf16]  — a lane's MFMA operand (row = lane & 31, 8 k of
// k-group lane >> 5) is one 16-byte ds_read_b128, consecutive lanes read consecutive 16-byte slots (conflict-free), and the
// same image serves K-contiguous and M/N-contiguous operands (the transposition happens in the registers of the loader:
// a K-contiguous operand gives each thread 8 consecutive k of one row, an M/N-contiguous one 8 rows-of-k of one column).
// The two k-groups of a plane are 4096 + 128 bytes apart so that the loader's writes (lane pairs alternate k-groups for
// K-contiguous operands) are conflict-free too.  LDS: 2 stages x 2 operands x 3 planes x 8448 B = 99 KB, one block per CU.
// The transposed-A form (dW) is split over K into slabs summed in a fixed order by a second kernel (no float atomics).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 16, NT = 512;
constexpr int TM = 4, TN = 2, WAVES_N = 4;
constexpr int KG_STRIDE = 256 * 8 + 64;  // bf16 elements between the two k-groups of a plane (4096 + 128 bytes)
constexpr int PLANE = 2 * KG_STRIDE;
constexpr int OPER = 3 * PLANE;
constexpr int STAGE = 2 * OPER;  // A planes then B planes

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base, const char* end) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const int64_t left = end - base;
  const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(left < 0 ? 0 : (left > 0xffffffffLL ? 0xffffffffLL : left)));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// two fp32 -> packed bf16 (round to nearest even; v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// x[0..7] -> the three bf16 planes (8 values each, one 16-byte vector per plane)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float a = x[2 * w], b = x[2 * w + 1];
    const uint32_t ph = pack_bf16(a, b);
    const float a1 = a - __uint_as_float(ph << 16), b1 = b - __uint_as_float(ph & 0xffff0000u);  // exact
    const uint32_t pm = pack_bf16(a1, b1);
    const float a2 = a1 - __uint_as_float(pm << 16), b2 = b1 - __uint_as_float(pm & 0xffff0000u);  // exact
    hi[w] = ph;
    mid[w] = pm;
    lo[w] = pack_bf16(a2, b2);
  }
}

// One operand side of the block tile.  KC: stored [rows][K] (k contiguous); otherwise stored [K][rows].
template <bool KC>
struct Loader {
  uint32_t voff;     // loop-invariant byte offset of this thread's first element from the tile origin at k = 0
  uint32_t row_step; // MC: bytes between consecutive k (ld * 4)
  int lds_off;       // bf16 element offset of this thread's 16-byte slot inside a plane
  int kg;            // k-group (0 / 1) this thread fills

  __device__ __forceinline__ void init(int tid, int64_t ld, int64_t rows, int64_t r0) {
    if constexpr (KC) {
      const int row = tid >> 1;
      kg = tid & 1;
      const int64_t rel = min((int64_t)row, rows - 1 - r0);  // clamped rows are never stored
      voff = (uint32_t)((rel * ld + 8 * kg) * 4);
      lds_off = kg * KG_STRIDE + row * 8;
      row_step = 0;
    } else {
      const int c = tid & 255;
      kg = tid >> 8;
      const int64_t rel = min((int64_t)c, rows - 1 - r0);
      voff = (uint32_t)(((int64_t)8 * kg * ld + rel) * 4);
      lds_off = kg * KG_STRIDE + c * 8;
      row_step = (uint32_t)(ld * 4);
    }
  }

  __device__ __forceinline__ void load(float (&s)[8], __amdgpu_buffer_rsrc_t r) const {
    if constexpr (KC) {
      const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
      const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i] = __uint_as_float(v0[i]);
        s[4 + i] = __uint_as_float(v1[i]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, e * row_step, 0));
    }
  }

  // zero the elements whose k lies at or beyond k_end (last, partial K step only)
  __device__ __forceinline__ void mask(float (&s)[8], int64_t k0, int64_t k_end) const {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (k0 + 8 * kg + e >= k_end) s[e] = 0.f;
  }

  __device__ __forceinline__ void store(const float (&s)[8], uint16_t* oper) const {
    u32x4 hi, mid, lo;
    split8(s, hi, mid, lo);
    *reinterpret_cast<u32x4*>(oper + lds_off) = hi;
    *reinterpret_cast<u32x4*>(oper + PLANE + lds_off) = mid;
    *reinterpret_cast<u32x4*>(oper + 2 * PLANE + lds_off) = lo;
  }
};

// TA: A stored [K][M]; TB: B stored [N][K].
template <bool TA, bool TB>
__global__ __launch_bounds__(NT, 2) void gemm_f32x3_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
                                                           int accumulate, int64_t k_chunk, float* __restrict__ slabs, int tiles_n,
                                                           int n_tiles) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * STAGE];

  // XCD-aware bijective remap (block b runs on XCD b % 8): every XCD gets a contiguous run of logical tiles
  const int bid = blockIdx.x;
  const int q = n_tiles / 8, rr = n_tiles % 8, xcd = bid % 8;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + bid / 8;
  const int64_t m0 = (int64_t)(logical / tiles_n) * BM;
  const int64_t n0 = (int64_t)(logical % tiles_n) * BN;
  const int64_t k_begin = (int64_t)blockIdx.z * k_chunk;
  const int64_t k_end = min(K, k_begin + k_chunk);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int i32 = lane & 31, h = lane >> 5;
  const int a_span = wm * (TM * 32), b_span = wn * (TN * 32);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int y = 0; y < TN; ++y) acc[x][y] = f32x16(0.f);

  Loader<!TA> la;
  Loader<TB> lb;
  la.init(tid, lda, M, m0);
  lb.init(tid, ldb, N, n0);
  const char* const a_end = reinterpret_cast<const char*>(A + (TA ? (K - 1) * lda + M : (M - 1) * lda + K));
  const char* const b_end = reinterpret_cast<const char*>(B + (TB ? (N - 1) * ldb + K : (K - 1) * ldb + N));
  const char* const a_origin = reinterpret_cast<const char*>(A + (TA ? m0 : m0 * lda));
  const char* const b_origin = reinterpret_cast<const char*>(B + (TB ? n0 * ldb : n0));
  const int64_t a_kstride = (TA ? lda : 1) * 4, b_kstride = (TB ? 1 : ldb) * 4;  // bytes per unit of k

  const int64_t n_steps = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;
  // Round 6 (the kernel stood at 0.46 of the bf16 peak for three rounds): a step used to be  fragment reads -> wait -> 48 MFMAs ->
  // wait for the next tile's loads -> ~100 vector instructions of splitting -> LDS stores -> barrier, one phase after the other in every
  // wave.  Now the operand tiles travel TWO steps ahead in registers (s?[parity]): step t requests tile t + 2, and splits and stores tile
  // t + 1 — loaded a whole step ago, so nothing waits for memory — in the issue slots between its own MFMAs (three vector instructions and
  // an LDS write behind every MFMA, pinned by sched_group_barrier).  The stage being written (t + 1) was last read in step t - 1, a
  // barrier ago.
  float sa[2][8] = {}, sb[2][8] = {};  // (zeroed: a range of one step splits and stores the never-loaded second set into the idle stage)
  auto tile_rsrc_a = [&](int64_t k0) __attribute__((always_inline)) { return make_rsrc(a_origin + k0 * a_kstride, a_end); };
  auto tile_rsrc_b = [&](int64_t k0) __attribute__((always_inline)) { return make_rsrc(b_origin + k0 * b_kstride, b_end); };
  if (n_steps > 0) {
    la.load(sa[0], tile_rsrc_a(k_begin));
    lb.load(sb[0], tile_rsrc_b(k_begin));
    if (k_begin + BK > k_end) {
      la.mask(sa[0], k_begin, k_end);
      lb.mask(sb[0], k_begin, k_end);
    }
    la.store(sa[0], lds);
    lb.store(sb[0], lds + OPER);
  }
  if (n_steps > 1) {  // tile 1 waits in the registers for step 0
    la.load(sa[1], tile_rsrc_a(k_begin + BK));
    lb.load(sb[1], tile_rsrc_b(k_begin + BK));
  }
  __syncthreads();

  auto step = [&](auto par_c, int64_t t) __attribute__((always_inline)) {
    constexpr int cur = decltype(par_c)::value, nxt = cur ^ 1;
    const bool has1 = t + 1 < n_steps, has2 = t + 2 < n_steps;
    const int64_t k1 = k_begin + (t + 1) * BK, k2 = k1 + BK;
    const uint16_t* a = lds + cur * STAGE + h * KG_STRIDE + (a_span + i32) * 8;
    const uint16_t* b = lds + cur * STAGE + OPER + h * KG_STRIDE + (b_span + i32) * 8;
    bf16x8_t fb[3][TN], fa[TM];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int y = 0; y < TN; ++y) fb[p][y] = *reinterpret_cast<const bf16x8_t*>(b + p * PLANE + y * 256);
    // tile t + 1 (in s?[nxt] since the previous step): masked if it is the partial last one, split and stored below, between the MFMAs
    if (__builtin_expect(has1 && k1 + BK > k_end, 0)) {  // the partial last tile of a K range whose length is no multiple of 16
      asm volatile("; partial tile" ::: "memory");         // (keeps this a branch: if-converted, its 16 64-bit compares + selects ran in every step)
      la.mask(sa[nxt], k1, k_end);
      lb.mask(sb[nxt], k1, k_end);
    }
    // A plane h: x {B_h, B_m, B_l};  A plane m: x {B_h, B_m};  A plane l: x {B_h}
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int x = 0; x < TM; ++x) fa[x] = *reinterpret_cast<const bf16x8_t*>(a + p * PLANE + x * 256);
      // unconditional (beyond the last tile the registers hold an old tile and the idle stage is never read again): as `if (has1)`
      // the split sat in a block of its own IN FRONT of the MFMAs instead of between them
      if (p == 0) la.store(sa[nxt], lds + nxt * STAGE);
      if (p == 1) lb.store(sb[nxt], lds + nxt * STAGE + OPER);
#pragma unroll
      for (int pb = 0; pb < 3 - p; ++pb)
#pragma unroll
        for (int x = 0; x < TM; ++x)
#pragma unroll
          for (int y = 0; y < TN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[x], fb[pb][y], acc[x][y], 0, 0, 0);
      // issue order inside the plane: every MFMA is followed by up to three vector instructions of the split, then LDS traffic
      constexpr int n_mfma = (3 - 0) * TM * TN;  // at most (plane h)
#pragma unroll
      for (int i = 0; i < n_mfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // tile t + 2 into the registers tile t just left (its loads fly through the whole next step)
    if (has2) {
      la.load(sa[cur], tile_rsrc_a(k2));
      lb.load(sb[cur], tile_rsrc_b(k2));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // this wave's LDS writes of tile t + 1 are done; no vmcnt(0): tile t + 2 stays in flight
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int64_t t = 0;
    for (; t + 1 < n_steps; t += 2) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < n_steps) step(std::integral_constant<int, 0>{}, t);
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* out = C;
  int64_t ldo = ldc;
  bool add = accumulate != 0;
  if (slabs) {
    out = slabs + (int64_t)blockIdx.z * M * N;
    ldo = N;
    add = false;
  }
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + a_span + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= M) continue;
#pragma unroll
      for (int y = 0; y < TN; ++y) {
        const int64_t col = n0 + b_span + y * 32 + i32;
        if (col >= N) continue;
        float* p = out + row * ldo + col;
        *p = add ? (*p + acc[x][y][r]) : acc[x][y][r];
      }
    }
}

// C = (accumulate ? C : 0) + sum_z slabs[z]   (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void x3_splitk_reduce_kernel(int64_t M, int64_t N, int S, const float* __restrict__ slabs,
                                                               float* __restrict__ C, int64_t ldc, int accumulate) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / N, col = i % N;
    float* p = C + row * ldc + col;
    float s = accumulate ? *p : 0.f;
    {
      int z = 0;
      for (; z + 8 <= S; z += 8) {  // eight partial values in flight, added in order (a plain loop is S dependent round trips)
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = slabs[(int64_t)(z + u) * total + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t8[u];
      }
      for (; z < S; ++z) s += slabs[(int64_t)z * total + i];
    }
    *p = s;
  }
}

struct Plan {
  int tiles_n, n_tiles, S;
  int64_t k_chunk;
};

Plan make_plan(int64_t M, int64_t N, int64_t K) {
  Plan p;
  p.tiles_n = (int)dh::ceil_div(N, BN);
  p.n_tiles = (int)dh::ceil_div(M, BM) * p.tiles_n;
  p.S = 1;
  p.k_chunk = dh::ceil_div(K > 0 ? K : 1, BK) * BK;
  // few output tiles and a long K (dW = X^T dZ): split K so that the grid is close to a whole number of rounds of the 256 CUs
  if (p.n_tiles < 512 && K >= 8192) {
    int64_t want = 1024 / p.n_tiles;
    if (want < 1) want = 1;
    int64_t max_s = K / (128 * BK);
    if (max_s < 1) max_s = 1;
    const int64_t S = want < max_s ? want : max_s;
    if (S > 1) {
      p.k_chunk = dh::ceil_div(dh::ceil_div(K, S), BK) * BK;
      p.S = (int)dh::ceil_div(K, p.k_chunk);
    }
  }
  return p;
}

// the split kernel wants enough 256 x 256 tiles to fill the chip and 16-byte aligned K-contiguous operands
bool x3_applies(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B, int64_t ldb) {
  if (K < 64 || lda >= ((int64_t)1 << 24) || ldb >= ((int64_t)1 << 24)) return false;
  if (dh::skinny_applies(M, N, K, trans_a)) return false;
  if (!trans_a && !(dh::aligned16(A) && lda % 4 == 0 && K % 4 == 0)) return false;
  if (trans_b && !(dh::aligned16(B) && ldb % 4 == 0 && K % 4 == 0)) return false;
  const Plan p = make_plan(M, N, K);
  return (int64_t)p.n_tiles * p.S >= 256;
}

}  // namespace

extern "C" size_t dh_gemm_f32x3_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan p = make_plan(M, N, K);
  const size_t own = p.S > 1 ? (size_t)p.S * (size_t)M * (size_t)N * sizeof(float) : 0;
  const size_t exact = dh_gemm_f32_workspace_bytes(M, N, K, trans_a, trans_b);  // the fallback's need
  return own > exact ? own : exact;
}

extern "C" int dh_gemm_f32x3(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda, const float* B,
                             int64_t ldb, float* C, int64_t ldc, int accumulate, void* workspace, size_t workspace_bytes,
                             dh_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32x3: negative size");
  if (M == 0 || N == 0) return DH_OK;
  if (!C || (K > 0 && (!A || !B))) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32x3: null operand");
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) return dh::fail(DH_ERR_INVALID, "dh_gemm_f32x3: leading dimension too small");
  if (!x3_applies(M, N, K, trans_a, trans_b, A, lda, B, ldb))  // small / unaligned problems: the exact-fp32 matrix-core kernel
    return dh_gemm_f32(M, N, K, trans_a, trans_b, A, lda, B, ldb, C, ldc, accumulate, workspace, workspace_bytes, stream);
  hipStream_t st = dh::as_stream(stream);
  const Plan p = make_plan(M, N, K);
  float* slabs = nullptr;
  if (p.S > 1) {
    const size_t need = (size_t)p.S * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need) return dh::fail(DH_ERR_WORKSPACE, "dh_gemm_f32x3: workspace %zu < %zu bytes", workspace_bytes, need);
    slabs = static_cast<float*>(workspace);
  }
  const dim3 grid((unsigned)p.n_tiles, 1, (unsigned)p.S);
#define DH_X3_LAUNCH(TA, TB)                                                                                                          \
  hipLaunchKernelGGL((gemm_f32x3_kernel<TA, TB>), grid, dim3(NT), 0, st, M, N, K, A, lda, B, ldb, C, ldc, accumulate, p.k_chunk, slabs, \
                     p.tiles_n, p.n_tiles)
  switch ((trans_a ? 2 : 0) | (trans_b ? 1 : 0)) {
    case 0: DH_X3_LAUNCH(false, false); break;
    case 1: DH_X3_LAUNCH(false, true); break;
    case 2: DH_X3_LAUNCH(true, false); break;
    default: DH_X3_LAUNCH(true, true); break;
  }
#undef DH_X3_LAUNCH
  int rc = dh::check_launch("dh_gemm_f32x3");
  if (rc != DH_OK) return rc;
  if (p.S > 1) {
    const int64_t total = M * N;
    const unsigned rgrid = (unsigned)(dh::ceil_div(total, 256) < 4096 ? dh::ceil_div(total, 256) : 4096);
    hipLaunchKernelGGL(x3_splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, st, M, N, p.S, slabs, C, ldc, accumulate);
    rc = dh::check_launch("dh_gemm_f32x3(split-K reduce)");
  }
  return rc;
}
