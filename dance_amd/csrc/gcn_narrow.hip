// The narrow GCN layer as two fused kernels (BASELINE config 5: SpaGCN's GraphConvolution 50 -> 50,
// spagcn.py:357-363 ``spmm(adj, mm(input, weight)) + bias``; also any layer with in, out <= 64).
//
// At these widths a row is 200 bytes: the transform-then-aggregate chain of the wide layer (skinny GEMM, SpMM, and in backward
// colsum, SpMM, split-K GEMM) is five launches of latency-bound kernels that together moved 1.85 GB of algorithmic traffic in
// 2.40 ms (0.10 of the HBM roofline, round 2).  Here the layer is evaluated AGGREGATE-FIRST — (A X) W instead of A (X W): the same
// function; fp32 rounding differs in the last bits, parity bar 1e-4 as for every layer (stated in DESIGN.md §3, SURVEY.md §8d
// "D = F only if the builder chooses aggregate-first") — which makes the forward ONE gather kernel and the weight gradient ONE
// streaming kernel:
//
//   forward : agg_i = sum_e a_e X[col_e, :]   (16 lanes x float4 per row, 4 rows per wavefront, 8 neighbours in flight per lane)
//             y_i   = act(agg_i W + b)        (32-row tiles of agg in LDS times W in LDS on the fp32 matrix cores)
//             agg is kept ([N, 64] fp32, column 63 = 1) for the backward.
//   backward: [dW; db] = [agg | 1]^T (dY * [y > 0])   one pass over agg and dY on the fp32 matrix cores (v_mfma_f32_32x32x2_f32,
//             K = rows: every wavefront owns a contiguous run of rows and a 64 x 64 accumulator), partial tiles reduced in fixed
//             order: deterministic.  The ones column makes db row 63 of the same product.
//   (dX = A^T ((dY * mask) W^T), needed only when the layer's input requires a gradient, stays on the generic kernels.)
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NW = 64;        // padded width of agg / W rows
constexpr int LDW = NW + 4;   // LDS row stride of W (float4-aligned, rows 4 banks apart)

constexpr int LDT = NW + 4;   // row stride of a wavefront's aggregated tile (float4-aligned, rows 4 banks apart)
constexpr int TROWS = 32;     // rows per wavefront tile = M of v_mfma_f32_32x32x2_f32

// VEC = 4: 16 lanes per row (X rows 16-byte aligned); VEC = 2: 32 lanes per row (8-byte aligned rows, e.g. ld = 50).
// Every wavefront owns tiles of 32 consecutive rows.  Phase 1 gathers: a lane group accumulates one row of A X in registers (up to
// eight neighbour rows in flight per lane; the column indices of the next row are fetched while this one is gathered) and parks it
// in the wavefront's LDS tile (and in AGG).  Phase 2 multiplies the 32 x 64 tile by W on the fp32 matrix cores: the round-4 kernel
// did that product with one ds_bpermute + one LDS read of W per (row, feature) — 0.62 ms at 500k rows x 50 -> 50, LDS-issue bound,
// four times what the gather itself costs.  K is walked as f = h * KS + s (h = lane half, s = k-step), so that a lane reads its A
// operands as KS consecutive floats of its tile row.  Nothing synchronises the block after W is staged.
template <int VEC>
__global__ __launch_bounds__(256) void gcn_narrow_forward_kernel(int64_t n_rows, int F, int H, const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                 const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                                                 const float* __restrict__ bias, int act, float* __restrict__ AGG,
                                                                 float* __restrict__ Y, int64_t ldy) {
  constexpr int G = NW / VEC;            // lanes per row
  constexpr int RW = 64 / G;             // rows per wavefront per gather step
  constexpr int NB = 8;                  // neighbour rows in flight per lane
  using V = typename std::conditional<VEC == 4, f32x4, f32x2>::type;
  __shared__ __attribute__((aligned(16))) float Ws[NW * LDW];
  __shared__ __attribute__((aligned(16))) float Ts[4 * TROWS * LDT];
  for (int i = threadIdx.x; i < NW * LDW; i += 256) {
    const int f = i / LDW, j = i - f * LDW;
    Ws[i] = (f < F && j < H) ? W[(int64_t)f * ldw + j] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane % G, sub = lane / G;
  const int c0 = g * VEC;
  const int i32 = lane & 31, h = lane >> 5;
  const int KS = (((F + 1) >> 1) + 3) & ~3;  // k-steps of the product (2 features each), a multiple of 4: <= 32 since F < 64
  const bool two = H > 32;                   // the second 32-column half of Y exists
  float* T = Ts + wave * TROWS * LDT;
  const float b0 = (bias && i32 < H) ? bias[i32] : 0.f;
  const float b1 = (bias && i32 + 32 < H) ? bias[i32 + 32] : 0.f;
  // Branch-free on purpose: with the load inside `if (c0 < F)` / `if (k + u < cnt)` the compiler closed every exec-masked region
  // with s_waitcnt vmcnt(0), so a row's neighbours were fetched ONE AT A TIME (0.37 ms at 500k rows even with every gather an L1
  // hit).  Every lane loads from a valid address (column 0 for lanes beyond the row's width, the group's clamped neighbour for
  // u >= cnt) and what must not count is zeroed by selects afterwards (a select does not propagate the NaN of a padding column).
  const int cl = c0 < F ? c0 : 0;
  auto xload = [&](int ck) -> V { return *reinterpret_cast<const V*>(X + (int64_t)ck * ldx + cl); };
  auto xmask = [&](V v, bool ok) -> V {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = (ok && c0 + i < F) ? v[i] : 0.f;
    return v;
  };
  // Workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD has its own L2.  Each XCD therefore walks ONE contiguous
  // eighth of the rows, its resident blocks side by side: on graphs whose neighbours are near in row order (spatial kNN in grid
  // order, a locality-ordered cell graph) the X rows the XCD's resident tiles gather are a sliding window that fits its L2, instead
  // of eight interleaved copies of the whole of X.
  const int64_t n_tiles = (n_rows + TROWS - 1) / TROWS;
  int64_t tile_begin = (int64_t)blockIdx.x * 4 + wave, tile_end = n_tiles, tile_step = (int64_t)gridDim.x * 4;
  if ((gridDim.x & 7) == 0) {
    const int64_t per_xcd = (((n_tiles + 7) >> 3) + 3) & ~(int64_t)3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tile_begin = xcd * per_xcd + slot * 4 + wave;
    tile_end = min(n_tiles, (xcd + 1) * per_xcd);
    tile_step = (int64_t)(gridDim.x >> 3) * 4;
  }
  for (int64_t tile = tile_begin; tile < tile_end; tile += tile_step) {
    const int64_t r0 = tile * TROWS;
    const int rp = rowptr[min(r0 + lane, n_rows)];  // lanes 0..32: the tile's row pointers (rows past the end are empty)
    int s = __shfl(rp, sub), t = __shfl(rp, sub + 1);
    int c = 0;
    float w = 0.f;
    if (s + g < t) {
      c = col[s + g];
      w = val ? val[s + g] : 1.f;
    }
    for (int it = 0; it < TROWS / RW; ++it) {
      const int ridx = it * RW + sub;
      int s2 = 0, t2 = 0, c2 = 0;
      float w2 = 0.f;
      if (it + 1 < TROWS / RW) {  // uniform
        s2 = __shfl(rp, ridx + RW);
        t2 = __shfl(rp, ridx + RW + 1);
        if (s2 + g < t2) {
          c2 = col[s2 + g];
          w2 = val ? val[s2 + g] : 1.f;
        }
      }
      V acc = V(0.f);
      for (int base = s; base < t; base += G) {
        if (base != s) {  // a row with more edges than the group has lanes
          const int e = base + g;
          c = 0;
          w = 0.f;
          if (e < t) {
            c = col[e];
            w = val ? val[e] : 1.f;
          }
        }
        const int cnt = min(G, t - base);
        for (int k = 0; k < cnt; k += NB) {
          int ck[NB];
          float wk[NB];
          V z[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            ck[u] = __shfl(c, min(k + u, G - 1), G);
            wk[u] = __shfl(w, min(k + u, G - 1), G);
          }
#pragma unroll
          for (int u = 0; u < NB; ++u) z[u] = xload(ck[u]);
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const V zu = xmask(z[u], k + u < cnt);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wk[u], zu[i], acc[i]);
          }
        }
      }
      *reinterpret_cast<V*>(T + ridx * LDT + c0) = acc;
      const int64_t row = r0 + ridx;
      if (AGG && row < n_rows) {  // [n_rows, 64]: the aggregated row, zero padded, with a 1 in column 63 (the bias row of the backward)
        V a = acc;
        if (c0 + VEC == NW) a[VEC - 1] = 1.f;
        *reinterpret_cast<V*>(AGG + row * NW + c0) = a;
      }
      s = s2, t = t2, c = c2, w = w2;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // Y tile (32 x 64) = T (32 x 2 KS) W (2 KS x 64).  A operand: lane (i32, h) holds T[i32][h KS + s]; B operand: W[h KS + s][i32 (+ 32)]
    f32x16 y0, y1;
#pragma unroll
    for (int r = 0; r < 16; ++r) y0[r] = 0.f, y1[r] = 0.f;
    const float* ta = T + i32 * LDT + h * KS;
    const float* wb = Ws + h * KS * LDW + i32;
    for (int s4 = 0; s4 < KS; s4 += 4) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ta + s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], wb[(s4 + e) * LDW], y0, 0, 0, 0);
        if (two) y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], wb[(s4 + e) * LDW + 32], y1, 0, 0, 0);
      }
    }
    // C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < n_rows) {
        if (i32 < H) {
          float v = y0[r] + b0;
          if (act == DH_ACT_RELU) v = fmaxf(v, 0.f);
          Y[row * ldy + i32] = v;
        }
        if (i32 + 32 < H) {
          float v = y1[r] + b1;
          if (act == DH_ACT_RELU) v = fmaxf(v, 0.f);
          Y[row * ldy + i32 + 32] = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next tile's rows overwrite T
  }
}

// partial[b] (64 x 64) = sum over the block's rows of agg_r^T g_r,  g = dY (* [y > 0]); see the file header
__global__ __launch_bounds__(256) void gcn_narrow_backward_kernel(int64_t n_rows, int H, int64_t rows_per_wave, const float* __restrict__ AGG,
                                                                  const float* __restrict__ dY, int64_t ldd, const float* __restrict__ Yact,
                                                                  int64_t ldy, float* __restrict__ partial) {
  __shared__ float red[NW * NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t w_global = (int64_t)blockIdx.x * 4 + wave;
  const int64_t r_begin = w_global * rows_per_wave;
  const int64_t r_end = min(n_rows, r_begin + rows_per_wave);
  f32x16 c[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[a][b][r] = 0.f;
  const bool j1 = i32 + 32 < H, j0 = i32 < H;
  auto gload = [&](int64_t r, int j, bool ok) -> float {
    if (r >= r_end || !ok) return 0.f;
    float v = dY[r * ldd + j];
    if (Yact && !(Yact[r * ldy + j] > 0.f)) v = 0.f;
    return v;
  };
  constexpr int U = 8;  // k-steps (of 2 rows) in flight
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * U) {
    float a0[U], a1[U], g0[U], g1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + 2 * u + h;
      const bool in = r < r_end;
      a0[u] = in ? AGG[r * NW + i32] : 0.f;
      a1[u] = in ? AGG[r * NW + 32 + i32] : 0.f;
      g0[u] = gload(r, i32, j0);
      g1[u] = gload(r, i32 + 32, j1);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], g0[u], c[0][0], 0, 0, 0);
      c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], g1[u], c[0][1], 0, 0, 0);
      c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], g0[u], c[1][0], 0, 0, 0);
      c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], g1[u], c[1][1], 0, 0, 0);
    }
  }
  // C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Waves 1..3 park their tile in
  // LDS, wave 0 adds them in order and writes the block's partial: fixed summation order.
  auto at = [&](int a, int b, int r) -> int { return (32 * a + (r & 3) + 8 * (r >> 2) + 4 * h) * NW + 32 * b + i32; };
  // waves 1, 2, 3 hand their tile to wave 0 through ONE 16 KB LDS tile, in that order (fixed summation order; 16 KB instead of
  // 48 KB keeps ten blocks resident per CU, the first version's three left a half-empty second round of blocks)
  for (int src = 1; src < 4; ++src) {
    if (wave == src) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[at(a, b, r)] = c[a][b][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) c[a][b][r] += red[at(a, b, r)];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* out = partial + (int64_t)blockIdx.x * NW * NW;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[at(a, b, r)] = c[a][b][r];
  }
}

// Two-level, fixed-order reduction of the per-block tiles: stage 1 sums runs of RCHUNK tiles (grid.y runs in parallel — one thread
// looping over all 1024 tiles was latency-bound at 0.13 ms, as long as the product itself), stage 2 sums the runs and scatters
// dW / db.
constexpr int RCHUNK = 32;

__global__ __launch_bounds__(256) void gcn_narrow_reduce1_kernel(int n_blocks, const float* __restrict__ partial, float* __restrict__ runs) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * RCHUNK, b1 = min(n_blocks, b0 + RCHUNK);
  float v[RCHUNK];
#pragma unroll
  for (int i = 0; i < RCHUNK; ++i) v[i] = (b0 + i < b1) ? partial[(int64_t)(b0 + i) * NW * NW + p] : 0.f;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < RCHUNK; ++i) acc += v[i];
  runs[(int64_t)blockIdx.y * NW * NW + p] = acc;
}

__global__ __launch_bounds__(256) void gcn_narrow_reduce2_kernel(int n_runs, int F, int H, const float* __restrict__ runs, float* __restrict__ dW,
                                                                 int64_t ldw, float* __restrict__ db) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int f = p / NW, j = p - f * NW;
  const bool is_w = f < F && j < H, is_b = db && f == NW - 1 && j < H;
  if (!is_w && !is_b) return;
  float acc = 0.f;
  {
      int b = 0;
      for (; b + 8 <= n_runs; b += 8) {  // eight partial values in flight, added in order (a plain loop is n_runs dependent round trips)
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = runs[(int64_t)(b + u) * NW * NW + p];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t8[u];
      }
      for (; b < n_runs; ++b) acc += runs[(int64_t)b * NW * NW + p];
    }
  if (is_w) dW[(int64_t)f * ldw + j] = acc;
  if (is_b) db[j] = acc;
}

// forward: a block is four wavefront tiles of 32 rows; 52 KB of LDS per block = three blocks per CU, the rest walk by grid stride
unsigned forward_blocks(int64_t n_rows) { return (unsigned)std::min<int64_t>((dh::ceil_div(n_rows, 4 * TROWS) + 7) & ~(int64_t)7, 3 * 256); }

int plan_blocks(int64_t n_rows) {  // one block per 2048 rows, at most 1024 blocks (16 MB of partial tiles)
  int64_t b = dh::ceil_div(n_rows, 2048);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" int dh_gcn_narrow_supported(int64_t in_features, int64_t out_features) {
  return in_features >= 1 && in_features < NW && out_features >= 1 && out_features <= NW;  // column 63 of agg carries the bias row
}

extern "C" int dh_gcn_narrow_forward_f32(int64_t n_rows, int64_t n_cols, int64_t in_features, int64_t out_features, const int32_t* rowptr,
                                         const int32_t* col, const float* val, const float* X, int64_t ldx, const float* W, int64_t ldw,
                                         const float* bias, int act, float* agg, float* Y, int64_t ldy, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: negative size");
  if (!dh_gcn_narrow_supported(in_features, out_features))
    return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: widths %lld -> %lld outside [1, 63] x [1, 64]", (long long)in_features, (long long)out_features);
  if (n_rows == 0) return DH_OK;
  if (!rowptr || !X || !W || !Y) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: null pointer");
  if (ldx < in_features || ldw < out_features || ldy < out_features) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: leading dimension too small");
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: bad act %d", act);
  if (agg && !dh::aligned16(agg)) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: agg must be 16-byte aligned");
  hipStream_t st = dh::as_stream(stream);
  // a float4 lane may read up to 3 floats past column F - 1 of a row: inside the row's stride when ldx >= round_up(F, 4)
  const bool v4 = dh::aligned16(X) && ldx % 4 == 0 && ldx >= ((in_features + 3) & ~(int64_t)3);
  const bool v2 = ((uintptr_t)X % 8 == 0) && ldx % 2 == 0 && ldx >= ((in_features + 1) & ~(int64_t)1);
  if (v4)
    hipLaunchKernelGGL(gcn_narrow_forward_kernel<4>, dim3(forward_blocks(n_rows)), dim3(256), 0, st, n_rows, (int)in_features,
                       (int)out_features, rowptr, col, val, X, ldx, W, ldw, bias, act, agg, Y, ldy);
  else if (v2)
    hipLaunchKernelGGL(gcn_narrow_forward_kernel<2>, dim3(forward_blocks(n_rows)), dim3(256), 0, st, n_rows, (int)in_features,
                       (int)out_features, rowptr, col, val, X, ldx, W, ldw, bias, act, agg, Y, ldy);
  else
    return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_forward_f32: X rows must be 8-byte aligned with an even leading dimension >= round_up(in, 2)");
  return dh::check_launch("dh_gcn_narrow_forward_f32");
}

extern "C" size_t dh_gcn_narrow_backward_workspace_bytes(int64_t n_rows) {
  if (n_rows <= 0) return 0;
  const int blocks = plan_blocks(n_rows);
  return (size_t)(blocks + dh::ceil_div(blocks, RCHUNK)) * NW * NW * sizeof(float);
}

extern "C" int dh_gcn_narrow_backward_f32(int64_t n_rows, int64_t in_features, int64_t out_features, const float* agg, const float* dY, int64_t ldd,
                                          const float* Y_act, int64_t ldy, float* dW, int64_t ldw, float* db, void* workspace,
                                          size_t workspace_bytes, dh_stream_t stream) {
  if (n_rows < 0) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_backward_f32: negative size");
  if (!dh_gcn_narrow_supported(in_features, out_features)) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_backward_f32: unsupported widths");
  if (!dW || ldw < out_features) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_backward_f32: bad dW");
  hipStream_t st = dh::as_stream(stream);
  if (n_rows == 0) {
    (void)dh::zero2d_async(dW, ldw * sizeof(float), out_features * sizeof(float), in_features, st);
    if (db) (void)dh::zero_async(db, out_features * sizeof(float), st);
    return DH_OK;
  }
  if (!agg || !dY || ldd < out_features || (Y_act && ldy < out_features)) return dh::fail(DH_ERR_INVALID, "dh_gcn_narrow_backward_f32: bad operand");
  const int blocks = plan_blocks(n_rows);
  const int runs = (int)dh::ceil_div(blocks, RCHUNK);
  if (!workspace || workspace_bytes < (size_t)(blocks + runs) * NW * NW * sizeof(float))
    return dh::fail(DH_ERR_WORKSPACE, "dh_gcn_narrow_backward_f32: workspace too small (dh_gcn_narrow_backward_workspace_bytes)");
  int64_t rows_per_wave = dh::ceil_div(n_rows, (int64_t)blocks * 4);
  rows_per_wave = (rows_per_wave + 1) & ~(int64_t)1;  // a k-step is two rows
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(gcn_narrow_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n_rows, (int)out_features, rows_per_wave, agg, dY, ldd,
                     Y_act, ldy, partial);
  int rc = dh::check_launch("dh_gcn_narrow_backward_f32");
  if (rc != DH_OK) return rc;
  float* run_sums = partial + (int64_t)blocks * NW * NW;
  hipLaunchKernelGGL(gcn_narrow_reduce1_kernel, dim3(NW * NW / 256, (unsigned)runs), dim3(256), 0, st, blocks, partial, run_sums);
  hipLaunchKernelGGL(gcn_narrow_reduce2_kernel, dim3(NW * NW / 256), dim3(256), 0, st, runs, (int)in_features, (int)out_features, run_sums, dW, ldw, db);
  return dh::check_launch("dh_gcn_narrow_backward_f32 (reduce)");
}
