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
#include <cstdlib>

#include "gemm_bf16_tile.h"

namespace {

using namespace dh_bf16;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

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
  int z = 0;
  for (; z + 8 <= slices; z += 8) {  // eight slab values in flight, added in slice order (one loop: `slices` dependent round trips, 30 us per launch)
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slabs[(int64_t)(z + u) * M * N + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; z < slices; ++z) v += slabs[(int64_t)z * M * N + i];
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

// ---- long-and-narrow products (the dense update of a layer: 1M x 400 -> 200) --------------------------------------------------
// C[M,N] = A[M,K] · B[N,K]^T with N <= 512 and K <= 512: the tiled kernel above spends such a product in prologues and epilogues
// (7 K-steps per tile, 36 % of the second column tile empty, A fetched once per column tile: 0.72 ms at 1M x 400 -> 200 for 1.2 GB
// = 0.21 of HBM).  Here B never moves: wave w of a workgroup keeps the fragments of column tile w (32 output columns x all of K:
// K / 16 x 4 registers) in registers for the whole kernel, and the workgroup streams row tiles of A through LDS — one coalesced
// 16-byte load per chunk, issued one tile ahead (registers, then a double-buffered LDS image whose 16-byte fragment reads are
// conflict-free: row stride = K-steps x 32 B + 16 B, an odd number of 16-byte slots).  The K loop has a compile-time trip count (KSM
// buckets; the image columns between K and 16 KSM are zero, as are the B fragments there), one barrier per row tile plus two for the
// output image; every byte of A and C crosses HBM once.  KSM = K-steps of 16 held in registers, MAXW = waves per workgroup (register
// budget 512 / (MAXW / 4)), RT = rows per tile.
//
// Measured (profiles/r03o_gemm_bf16_rows.json): 1M x 400 -> 200 with bias + ReLU 0.39 ms = 3.1 TB/s = 0.38 of HBM (tiled kernel: 0.72
// ms).  What is left: the B fragments take 104 of 222 registers, so a CU holds ONE workgroup and the three phases of a row tile run
// one after the other — alone, the loads take 0.14 ms (5.6 TB/s), the MFMAs + fragment reads 0.13 ms, the output 0.17 ms; together 0.39.
// Keeping reads in flight throughout was tried and is NOT the lever: a ring of four 32-row images filled by LDS DMA three tiles ahead
// (global_load_lds_dwordx4, no staging registers, counted vmcnt so that output stores never block the wait; parity-green) measured
// 0.42 ms — the MFMA + output phases of the one 8-wave workgroup run in lockstep behind their own LDS round trips and barriers (0.29
// ms without any load), and 32-row tiles double that per-tile cost.  The next step is more waves per CU: two waves per column tile,
// each holding half of K (52 registers of B instead of 104 -> four waves per SIMD), partial sums combined in the output image.
// Workgroup barrier that orders LDS traffic only (__syncthreads() is a workgroup-scope fence: the compiler may drain vmcnt in front of
// it, which would make every barrier of the tile loop wait for the output stores just issued and for the prefetch of the next tile).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KSM, int MAXW, int RT>
__global__ __launch_bounds__(64 * MAXW) void gemm_bf16_rows_kernel(int64_t M, int N, int kp, const uint16_t* __restrict__ A, int64_t lda,
                                                                   const uint16_t* __restrict__ B, int64_t ldb, void* __restrict__ C, int64_t ldc,
                                                                   Epilogue ep, int n_waves_cols, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_lds[];
  constexpr int MT = RT / 32;
  constexpr int MINW = MAXW >= 8 ? 8 : 4;                                 // the host launches at least this many waves
  constexpr int SCM = (RT * KSM * 2 + 64 * MINW - 1) / (64 * MINW) + 1;   // staging passes (rows are covered by whole groups of `cpr` threads)
  constexpr int STRIDE = KSM * 32 + 16;                                   // LDS bytes per row: an odd number of 16-byte slots
  constexpr int BUF = RT * STRIDE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int r32 = lane & 31, kh = lane >> 5;
  const int cpr = kp / 8;  // 16-byte chunks of a row that exist (kp is a multiple of 8)

  // B fragments of this wave's column tile: lane = (column r32, k half kh) holds B[n][16 s + 8 kh .. + 8]; zero beyond kp, so the
  // K loop below always runs its KSM steps without a branch (a guarded step cost a copy of every accumulator per step)
  bf16x8 bfrag[KSM];
  {
    const int n = wave * 32 + r32;
    const bool live_n = wave < n_waves_cols && n < N;
    const uint16_t* bp = B + (int64_t)(live_n ? n : 0) * ldb + 8 * kh;
#pragma unroll
    for (int s = 0; s < KSM; ++s) {
      u32x4 v = u32x4(0u);
      if (live_n && 16 * s + 8 * kh < kp) v = *reinterpret_cast<const u32x4*>(bp + 16 * s);
      bfrag[s] = __builtin_bit_cast(bf16x8, v);
    }
  }
  // both A images start as zeros: the chunks between kp and 16 KSM are never written again and multiply the zero B fragments
  for (int i = tid; i < 2 * BUF / 16; i += nthreads) reinterpret_cast<u32x4*>(rows_lds)[i] = u32x4(0u);
  __syncthreads();

  // staging: a pass covers rpp = nthreads / cpr whole rows, thread (r0, c0) takes chunk c0 of row r0 + pass * rpp
  const int rpp = nthreads / cpr;
  const int r0 = tid / cpr, c0 = tid - r0 * cpr;
  const bool stager = r0 < rpp;
  u32x4 stage[SCM];
  // The loads are unconditional — out-of-range rows / idle threads read a clamped (valid) address and the value is dropped or never
  // stored (behind a divergent guard every load sits in its own basic block with a full vmcnt(0) wait)
  const int c0c = stager ? c0 : 0;
  auto load_tile = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t m0 = tile * RT;
#pragma unroll
    for (int q = 0; q < SCM; ++q) {
      const int64_t m = min(m0 + r0 + q * rpp, M - 1);
      stage[q] = *reinterpret_cast<const u32x4*>(A + m * lda + c0c * 8);
    }
  };
  auto stage_to_lds = [&](int buf) __attribute__((always_inline)) {
    unsigned char* lp = rows_lds + buf * BUF + r0 * STRIDE + c0 * 16;
#pragma unroll
    for (int q = 0; q < SCM; ++q)
      if (stager && r0 + q * rpp < RT) *reinterpret_cast<u32x4*>(lp + q * rpp * STRIDE) = stage[q];
  };

  // Per row tile: barrier (image `buf` complete) -> issue the loads of the next tile -> MFMAs on this tile -> wait for the loads and
  // write them into the other image -> store this tile's C.  The output stores are issued AFTER the wait, so the only vmcnt wait of
  // an iteration never sits behind a store that was just issued (stores count in vmcnt on gfx9: waiting at the top of the loop
  // exposed the write latency of every tile).
  const int64_t n_tiles = (M + RT - 1) / RT;
  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const float bias_col = (ep.bias && wave * 32 + r32 < N) ? ep.bias[wave * 32 + r32] : 0.f;  // of this lane's accumulator column (loaded once)
  load_tile(tile);
  stage_to_lds(0);
  // Nothing may be pending when the loop is entered: with the bias (or a B fragment) still in flight the compiler put a counted vmcnt
  // wait in front of the MFMAs of EVERY iteration (ISA), and since vmcnt retires in order that wait also covers the output stores of the
  // previous tile.
  asm volatile("" ::"v"(bias_col));
#pragma unroll
  for (int s = 0; s < KSM; ++s) asm volatile("" ::"v"(bfrag[s]));
  int buf = 0;
  for (; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
    lds_barrier();  // image `buf` is complete; image `buf ^ 1` is no longer read by anyone
    const int64_t next = tile + gridDim.x;
    if (next < n_tiles) load_tile(next);
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    if (wave < n_waves_cols) {
      const unsigned char* img = rows_lds + buf * BUF + r32 * STRIDE + kh * 16;
      // fragments are read two K-steps ahead of their MFMAs (scheduling barriers keep the compiler from hoisting ALL reads of the tile)
      bf16x8 af[3][MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[0][m] = *reinterpret_cast<const bf16x8*>(img + m * 32 * STRIDE);
      if (KSM > 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) af[1][m] = *reinterpret_cast<const bf16x8*>(img + m * 32 * STRIDE + 32);
      }
#pragma unroll
      for (int s = 0; s < KSM; ++s) {
        if (s + 2 < KSM) {
#pragma unroll
          for (int m = 0; m < MT; ++m) af[(s + 2) % 3][m] = *reinterpret_cast<const bf16x8*>(img + m * 32 * STRIDE + (s + 2) * 32);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s % 3][m], bfrag[s], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (next < n_tiles) stage_to_lds(buf ^ 1);
    // Epilogue, 32 rows at a time, through ONE fp32 image [32][32 n_waves_cols] shared by the workgroup: every wave drops its accumulators
    // (+ bias, ReLU) at their (row, column), then all threads stream the image out in row-major order, 16 bytes per lane.  A wave's own
    // 32 columns are only 64 bytes of a bf16 row (half a line per store); streamed in row-major order the stores cover whole lines, and
    // bias / ReLU / rounding / accumulate need no per-element guards (element-wise stores from the accumulators made the compiler keep 32
    // guarded address computations alive across the tile loop and spill the B fragments).
    float* ct = reinterpret_cast<float*>(rows_lds + 2 * BUF);
    const int NC = 32 * n_waves_cols;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (wave < n_waves_cols) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = acc[m][i] + bias_col;
          if (ep.act == DH_ACT_RELU) v = fmaxf(v, 0.f);
          ct[((i & 3) + 8 * (i >> 2) + 4 * kh) * NC + wave * 32 + r32] = v;
        }
      }
      lds_barrier();
      const int64_t row0 = tile * RT + m * 32;
      const int rows_here = (int)min((int64_t)32, M - row0);
      if (rows_here > 0) {
        if (vec_ok) {  // N a multiple of the chunk, C and its rows 16-byte aligned
          const int per = ep.c_bf16 ? 8 : 4, cprc = N / per;
          for (int idx = tid; idx < rows_here * cprc; idx += nthreads) {
            const int row = idx / cprc, c = (idx - row * cprc) * per;
            const float* src = ct + row * NC + c;
            f32x4_t lo = *reinterpret_cast<const f32x4_t*>(src);
            if (ep.c_bf16) {
              f32x4_t hi = *reinterpret_cast<const f32x4_t*>(src + 4);
              uint16_t* q = static_cast<uint16_t*>(C) + (row0 + row) * ldc + c;
              if (ep.accumulate) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(q);
                lo[0] += __uint_as_float(o[0] << 16); lo[1] += __uint_as_float(o[0] & 0xffff0000u);
                lo[2] += __uint_as_float(o[1] << 16); lo[3] += __uint_as_float(o[1] & 0xffff0000u);
                hi[0] += __uint_as_float(o[2] << 16); hi[1] += __uint_as_float(o[2] & 0xffff0000u);
                hi[2] += __uint_as_float(o[3] << 16); hi[3] += __uint_as_float(o[3] & 0xffff0000u);
              }
              u32x4 o;
              o[0] = f32_to_bf16(lo[0]) | (f32_to_bf16(lo[1]) << 16);
              o[1] = f32_to_bf16(lo[2]) | (f32_to_bf16(lo[3]) << 16);
              o[2] = f32_to_bf16(hi[0]) | (f32_to_bf16(hi[1]) << 16);
              o[3] = f32_to_bf16(hi[2]) | (f32_to_bf16(hi[3]) << 16);
              *reinterpret_cast<u32x4*>(q) = o;
            } else {
              float* q = static_cast<float*>(C) + (row0 + row) * ldc + c;
              if (ep.accumulate) lo += *reinterpret_cast<const f32x4_t*>(q);
              *reinterpret_cast<f32x4_t*>(q) = lo;
            }
          }
        } else {
          for (int idx = tid; idx < rows_here * N; idx += nthreads) {
            const int row = idx / N, c = idx - row * N;
            float v = ct[row * NC + c];
            if (ep.c_bf16) {
              uint16_t* q = static_cast<uint16_t*>(C) + (row0 + row) * ldc + c;
              if (ep.accumulate) v += __uint_as_float((unsigned int)*q << 16);
              *q = (uint16_t)f32_to_bf16(v);
            } else {
              float* q = static_cast<float*>(C) + (row0 + row) * ldc + c;
              if (ep.accumulate) v += *q;
              *q = v;
            }
          }
        }
      }
      if (m + 1 < MT) lds_barrier();  // (after the last block the barrier at the top of the tile loop protects the image)
    }
  }
}

bool rows_applies(int64_t M, int64_t N, int64_t kp, int slices) {
  return slices == 1 && M >= 2048 && ((N <= 256 && kp <= 512) || (N <= 512 && kp <= 208));
}

template <int KSM, int MAXW, int RT>
int rows_launch(int64_t M, int64_t N, int64_t kp, const uint16_t* a, int64_t la, const uint16_t* b, int64_t lb, void* C, int64_t ldc, const Epilogue& ep,
                hipStream_t st) {
  constexpr int MINW = MAXW >= 8 ? 8 : 4;
  const int nwc = (int)((N + 31) / 32);
  const int waves = nwc < MINW ? MINW : nwc;  // the staging plan is sized for at least this many threads
  const size_t lds = (size_t)2 * RT * (KSM * 32 + 16) + (size_t)32 * 32 * nwc * sizeof(float);  // two A images + the fp32 output image of 32 rows
  const int vec_ok = dh::aligned16(C) && ldc % (ep.c_bf16 ? 8 : 4) == 0 && N % (ep.c_bf16 ? 8 : 4) == 0;
  static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_rows_kernel<KSM, MAXW, RT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  if (!lds_ok) return dh::fail(DH_ERR_LAUNCH, "dh_gemm_bf16: cannot raise the dynamic LDS limit");
  const int64_t n_tiles = dh::ceil_div(M, (int64_t)RT);
  const int per_cu = lds <= 76 * 1024 && waves <= 8 && KSM <= 4 ? 2 : 1;  // 2 workgroups per CU where LDS and registers (<= 128) allow
  const unsigned grid = (unsigned)(n_tiles < 256 * per_cu ? n_tiles : 256 * per_cu);
  hipLaunchKernelGGL((gemm_bf16_rows_kernel<KSM, MAXW, RT>), dim3(grid), dim3(64 * waves), lds, st, M, (int)N, (int)kp, a, la, b, lb, C, ldc, ep, nwc, vec_ok);
  return dh::check_launch("dh_gemm_bf16(rows)");
}

int rows_dispatch(int64_t M, int64_t N, int64_t kp, const uint16_t* a, int64_t la, const uint16_t* b, int64_t lb, void* C, int64_t ldc, const Epilogue& ep,
                  hipStream_t st) {
  const int ks = (int)((kp + 15) / 16);
  if (N > 256) {  // up to 16 column tiles: 1024 threads, 128 registers
    if (ks <= 4) return rows_launch<4, 16, 32>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
    if (ks <= 8) return rows_launch<8, 16, 32>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
    return rows_launch<13, 16, 32>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
  }
  if (ks <= 4) return rows_launch<4, 8, 64>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
  if (ks <= 8) return rows_launch<8, 8, 64>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
  if (ks <= 13) return rows_launch<13, 8, 64>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
  if (ks <= 16) return rows_launch<16, 8, 64>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
  if (ks <= 26) return rows_launch<26, 8, 64>(M, N, kp, a, la, b, lb, C, ldc, ep, st);
  return rows_launch<32, 8, 32>(M, N, kp, a, la, b, lb, C, ldc, ep, st);  // (two 64-row images of 1 KB rows + the patches exceed 160 KB)
}

// ---- dW = dY^T X with BOTH operands stored K-major (trans_a, row-major B) and a long K: C[M,N] = sum_k A[k][m] B[k][n] ----------------
// The generic path transposes both operands into K-contiguous copies first (two extra passes over 2 x (M + N) K bytes: 1.61 ms for
// the 200 x 400 gradient over 1M cells, of which the product itself is 0.15).  Here a workgroup streams 64-row slabs of A and B as
// they lie and TRANSPOSES THEM ON THE WAY INTO LDS: a thread loads an 8 (k) x 8 (columns) block — eight 16-byte loads, rows 8 apart
// for neighbouring lanes, eight lanes on one 128-byte line —, turns it in registers (32 byte permutes) and stores eight 16-byte runs
// of k for its eight columns, so that an MFMA fragment (8 consecutive k of one column) is ONE 16-byte LDS read.  LDS image
// [column][64 k + 8 pad] (144-byte rows: the 16 lanes of a quarter-wave read or write distinct banks).  8 waves, two per SIMD: waves
// 0 - 3 load A, waves 4 - 7 B; a wave owns all (<= 7) row tiles of C for one column tile of the workgroup's 256 columns.  The slabs of
// the next TWO steps travel in registers while the current one is multiplied (a streaming kernel with one workgroup per CU lives on
// the bytes it keeps in flight).  The K range is split over the workgroups, the column blocks of one K slice on one XCD (fp32 slabs,
// summed in slice order by gemm_bf16_reduce_kernel).  Shapes: M <= 224, M % 8 == 0, N % 8 == 0, 16-byte aligned rows.
// Measured at 200 x 400 x 1M: 0.405 ms = 3.0 TB/s (profiles/r04zp_gemm_bf16_tn.json).  The first form of this kernel left the slabs
// untransposed and gathered every fragment with 8 two-byte LDS reads: 0.55 - 0.57 ms, bound by the LDS instruction rate (~5 cycles per
// wave instruction and CU); fetching row pairs as dwords (ds_read2_b32 + permute) was slower still (0.65).
constexpr int TT_KC = 64;  // k rows per step
constexpr int TX_LD = 72;  // bf16 per LDS row: 64 k + 8 pad = 144 bytes
constexpr size_t TX_LDS = (size_t)2 * (224 + 256) * TX_LD * sizeof(uint16_t);  // 138 240 B

template <int MT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_bf16_tn_tall_kernel(int M, int N, int64_t K, const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ B,
                               int64_t ldb, float* __restrict__ slabs, int64_t k_per_slice, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) uint16_t tx_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, kb = lane >> 5;
  const int j8 = blockIdx.x >> 3;
  const int slice = (j8 / n_blocks) * 8 + (blockIdx.x & 7);
  const int n_blk = (j8 % n_blocks) * 256;
  const int64_t k_begin = (int64_t)slice * k_per_slice;
  if (k_begin >= K) return;
  const int k_len = (int)((k_begin + k_per_slice < K ? k_begin + k_per_slice : K) - k_begin);
  const int n_steps = (k_len + TT_KC - 1) / TT_KC;
  const int pa = M / 8, pb_all = N / 8;
  const int pb0 = n_blk / 8, pb = (pb_all - pb0) < 32 ? (pb_all - pb0) : 32;
  // loader role: octet o = lane & 7 (k rows 8 o .. 8 o + 7 of the slab), piece pc = 8 (wave & 3) + (lane >> 3); waves 0 - 3: A, 4 - 7: B
  const bool load_b = wave >= 4;
  const int o = lane & 7, pc = 8 * (wave & 3) + (lane >> 3);
  const int p_lim = load_b ? pb : pa;
  const bool stores = pc < (load_b ? 32 : 4 * MT);  // (pieces beyond the A image are loaded from a clamped address and dropped)
  const char* const g_base = load_b ? reinterpret_cast<const char*>(B + k_begin * ldb + (int64_t)pb0 * 8) : reinterpret_cast<const char*>(A + k_begin * lda);
  const unsigned ld_b = (unsigned)(load_b ? ldb : lda) * 2u;
  const unsigned col_b = (unsigned)min(pc, p_lim - 1) * 16u;
  auto load = [&](int k0, u32x4 (&rg)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned kr = (unsigned)min(k0 + 8 * o + j, k_len - 1);
      rg[j] = *reinterpret_cast<const u32x4*>(g_base + (kr * ld_b + col_b));
    }
  };
  const unsigned sel_lo = 0x05040100u, sel_hi = 0x07060302u;
  auto store = [&](int buf, int k0, const u32x4 (&rg)[8]) __attribute__((always_inline)) {
    uint16_t* img = tx_lds + (size_t)buf * (224 + 256) * TX_LD + (load_b ? 224 * TX_LD : 0) + (size_t)(8 * pc) * TX_LD + 8 * o;
    const int k_left = k_len - (k0 + 8 * o);  // rows of this octet inside the slice (<= 0: none)
    const bool col_live = pc < p_lim;
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // column 8 pc + c: its eight k values, two per dword
      u32x4 q;
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        unsigned lo = rg[2 * p2][c >> 1], hi = rg[2 * p2 + 1][c >> 1];
        if (2 * p2 >= k_left) lo = 0u;      // (only the last step of a slice has rows to blank)
        if (2 * p2 + 1 >= k_left) hi = 0u;
        q[p2] = __builtin_amdgcn_perm(hi, lo, (c & 1) ? sel_hi : sel_lo);
      }
      if (stores) *reinterpret_cast<u32x4*>(img + c * TX_LD) = col_live ? q : u32x4{0u, 0u, 0u, 0u};
    }
  };
  f32x16 tacc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) tacc[i][e] = 0.f;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const uint16_t* a_img = tx_lds + (size_t)buf * (224 + 256) * TX_LD;
    const uint16_t* b_img = a_img + 224 * TX_LD;
#pragma unroll
    for (int ks = 0; ks < TT_KC / 16; ++ks) {
      const int koff = 16 * ks + 8 * kb;
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b_img + (size_t)(wave * 32 + r) * TX_LD + koff);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        tacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(a_img + (size_t)(i * 32 + r) * TX_LD + koff), fb, tacc[i], 0, 0, 0);
    }
  };
  u32x4 rg0[8], rg1[8];
  load(0, rg0);
  store(0, 0, rg0);
  load(TT_KC, rg1);
  __syncthreads();
  for (int t = 0; t < n_steps; t += 2) {
    const int k0 = t * TT_KC;
    load(k0 + 2 * TT_KC, rg0);
    compute(0);
    store(1, k0 + TT_KC, rg1);
    __syncthreads();
    load(k0 + 3 * TT_KC, rg1);
    compute(1);
    store(0, k0 + 2 * TT_KC, rg0);
    __syncthreads();
  }
  float* slab = slabs + (int64_t)slice * M * N;
  const int n = n_blk + wave * 32 + r;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * kb;
      if (m < M && n < N) slab[(int64_t)m * N + n] = tacc[i][e];
    }
}

bool tn_tall_applies(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const void* A, int64_t lda, const void* B, int64_t ldb) {
  return trans_a && !trans_b && M <= 224 && M % 8 == 0 && N % 8 == 0 && N <= 4096 && K >= 64 * TT_KC && lda % 8 == 0 && ldb % 8 == 0 &&
         lda < (1 << 20) && ldb < (1 << 20) && dh::aligned16(A) && dh::aligned16(B);
}
struct TallPlan {
  int n_blocks, slices;
  int64_t k_per_slice;
  size_t slab_bytes;
};
TallPlan tn_tall_plan(int64_t M, int64_t N, int64_t K) {
  TallPlan p;
  p.n_blocks = (int)dh::ceil_div(N, 256);
  int64_t slices = 256 / p.n_blocks;  // one workgroup per CU
  const int64_t max_slices = K / (8 * TT_KC);
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  p.k_per_slice = dh::ceil_div(dh::ceil_div(K, slices), TT_KC) * TT_KC;
  p.slices = (int)dh::ceil_div(K, p.k_per_slice);
  p.slab_bytes = (size_t)p.slices * M * N * sizeof(float);
  return p;
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
  if (tn_tall_applies(M, N, K, trans_a, trans_b, A, lda, B, ldb)) {
    const TallPlan tp = tn_tall_plan(M, N, K);
    if (workspace && workspace_bytes >= tp.slab_bytes && dh::aligned16(workspace)) {  // (the size query covers it: the generic path needs more)
      float* slabs = static_cast<float*>(workspace);
      const unsigned grid = (unsigned)((tp.slices + 7) / 8 * 8 * tp.n_blocks);
#define DH_TT(MTV)                                                                                                                 \
  do {                                                                                                                             \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_tn_tall_kernel<MTV>),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)TX_LDS) == hipSuccess;             \
    if (!ok) return dh::fail(DH_ERR_LAUNCH, "dh_gemm_bf16: cannot raise the dynamic LDS limit");                                   \
    hipLaunchKernelGGL(gemm_bf16_tn_tall_kernel<MTV>, dim3(grid), dim3(512), TX_LDS, st, (int)M, (int)N, K, A, lda, B, ldb, slabs,  \
                       tp.k_per_slice, tp.n_blocks);                                                                               \
  } while (0)
      switch ((int)((M + 31) / 32)) {
        case 1: DH_TT(1); break;
        case 2: DH_TT(2); break;
        case 3: DH_TT(3); break;
        case 4: DH_TT(4); break;
        case 5: DH_TT(5); break;
        case 6: DH_TT(6); break;
        default: DH_TT(7); break;
      }
#undef DH_TT
      hipLaunchKernelGGL(gemm_bf16_reduce_kernel, dim3((unsigned)dh::ceil_div(M * N, 256)), dim3(256), 0, st, M, N, tp.slices, slabs, C, ldc, ep);
      return dh::check_launch("dh_gemm_bf16");
    }
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
  if (rows_applies(M, N, p.kp, p.slices))  // long and narrow: B stationary in registers, A streamed once
    return rows_dispatch(M, N, p.kp, a, la, b, lb, C, ldc, ep, st);
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
