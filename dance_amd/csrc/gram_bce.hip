// Fused inner-product decoder of graph-sc (dance/modules/single_modality/clustering/graphsc.py:208-216, :405-411):
//   adj_logits = z z^T (B x B), loss = norm * mean(binary_cross_entropy_with_logits(adj_logits, adj, pos_weight)).
// The target `adj` is zero except for the few edges among the batch's own cells, so the dense part of the loss is
//   sum_ij softplus(x_ij),  x = z z^T,  and its gradient  dz_i = 2 * sum_j sigmoid(x_ij) z_j   (x is symmetric);
// the y = 1 corrections live on the edge list and stay with the caller.  Unfused this is a B x B fp32 GEMM, two passes over
// the 268 MB logit matrix (B = 8192) and two more B x B x d GEMMs in the backward.  Here the logits never leave the
// registers (flash-attention shape):
//
//   dh_gram_sigmoid_f32:  rowloss[i] = sum_j softplus(<z_i, z_j>),   O[i, :] = sum_j sigmoid(<z_i, z_j>) z_j
//
// both products on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).  One wavefront owns 32 rows i: their features are
// the B operand of the first product and stay in registers for the whole kernel (DP / 2 VGPRs); the j rows stream through
// LDS in tiles of 32 (double buffered, one barrier per tile, shared by the four wavefronts of the workgroup).  The first
// product is computed transposed, S^T[j][i], so that its accumulator layout (column = lane % 32 = i) IS the A-operand layout
// of the second product (row = lane % 32 = i, k = the j of register r / lane half): sigmoid(S) feeds the second MFMA
// straight from the accumulator registers, no LDS round trip.  The j range is split across blockIdx.y to fill the chip; the
// partial sums are combined in a fixed order by a second kernel (deterministic, no atomics).
//
// Roofline: matrix-core bound, 4 * B^2 * DP flop (DP = d padded to 64): B = 8192, d = 300 -> 85.9 GFLOP = 0.55 ms at the
// 157.3 TFLOP/s fp32 peak; HBM traffic is the 10 MB of z (L2 resident) plus the partial outputs.
#include <type_traits>

#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

constexpr int BI = 128;  // rows i per workgroup (4 wavefronts x 32)
constexpr int BJ = 32;   // rows j per LDS tile
constexpr int MAX_D = 320;

__host__ __device__ constexpr int lds_stride(int dp) { return dp + 4; }  // == 4 (mod 64): the 32 rows of a tile start 4 banks apart

// DP: d padded to a multiple of 64.  VEC: ldz % 4 == 0, d % 4 == 0 and Z 16-byte aligned (float4 loads), else scalar loads.
// MODE: the pair (f, g = f') evaluated on every logit x = <z_i, z_j>: rowloss[i] = sum_j f(x_ij), O[i] = sum_j g(x_ij) z_j.
//   0: f = softplus, g = sigmoid                       graph-sc's BCE-with-logits against a (nearly) all-zero target
//   1: f = sigmoid^2, g = 2 sigmoid^2 (1 - sigmoid)    scTAG's MSE(sigmoid(z z^T), adj) (sctag.py:470-471, :254): the dense part
//                                                      sum_ij sigmoid(x_ij)^2 of sum_ij (sigmoid(x_ij) - a_ij)^2
template <int DP, bool VEC, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gram_sigmoid_kernel(int n_r, const float* __restrict__ Zr, int64_t ldr, int n, int d, const float* __restrict__ Z, int64_t ldz,
                         int j_per_split, int n_pad, float* __restrict__ Opart, double* __restrict__ Lpart) {
  constexpr int STRIDE = lds_stride(DP);
  constexpr int HALF = DP / 2;
  constexpr int NT = DP / 32;              // column tiles of the second product
  constexpr int ROW4 = DP / 4;             // float4 per tile row
  constexpr int F4 = BJ * ROW4 / 256;      // float4 per thread per tile (= DP / 32)
  extern __shared__ float lds[];           // [2][BJ][STRIDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * BI + wave * 32;
  const int split = blockIdx.y;
  const int j_begin = split * j_per_split;
  const int j_end = min(n, j_begin + j_per_split);
  const int n_tiles = (j_end - j_begin + BJ - 1) / BJ;

  // rows i come from Zr [n_r, d] (the square pass: Zr = Z), columns j from Z [n, d]
  auto load4_from = [&](const float* M, int64_t ld, int rows, int row, int col) -> f32x4 {  // M[row][col .. col + 3], zero outside the matrix
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const float* p = M + (int64_t)row * ld + col;
      if (VEC) {
        if (col < d) v = *reinterpret_cast<const f32x4*>(p);
      } else {
        if (col + 0 < d) v.x = p[0];
        if (col + 1 < d) v.y = p[1];
        if (col + 2 < d) v.z = p[2];
        if (col + 3 < d) v.w = p[3];
      }
    }
    return v;
  };
  auto load4 = [&](int row, int col) -> f32x4 { return load4_from(Z, ldz, n, row, col); };

  // B operand of the first product: lane (m, h) holds Z[i0 + m][h * HALF + t], t = 0 .. HALF - 1 (k is enumerated as
  // (h, t); the A operand read from LDS uses the same enumeration, so any order is a valid K order)
  float zi[HALF];
  static_for<HALF / 4>([&](auto q_c) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value;
    const f32x4 v = load4_from(Zr, ldr, n_r, i0 + m, h * HALF + 4 * q);
    zi[4 * q + 0] = v.x; zi[4 * q + 1] = v.y; zi[4 * q + 2] = v.z; zi[4 * q + 3] = v.w;
  });

  f32x4 st[F4];
  auto load_tile = [&](int j0) {
    static_for<F4>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      const int f = tid + 256 * q;
      const int row = f / ROW4, c4 = f - row * ROW4;
      const int j = j0 + row;
      st[q] = load4(j < j_end ? j : n, 4 * c4);
    });
  };
  auto store_tile = [&](int buf) {
    float* base = lds + buf * (BJ * STRIDE);
    static_for<F4>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      const int f = tid + 256 * q;
      const int row = f / ROW4, c4 = f - row * ROW4;
      *reinterpret_cast<f32x4*>(base + row * STRIDE + 4 * c4) = st[q];
    });
  };

  f32x16 O[NT];
  static_for<NT>([&](auto y_c) __attribute__((always_inline)) {
    constexpr int y = decltype(y_c)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[y][r] = 0.f;
  });
  double lsum = 0.0;

  if (n_tiles > 0) {
    load_tile(j_begin);
    store_tile(0);
  }
  __syncthreads();

  for (int t = 0; t < n_tiles; ++t) {
    const float* buf = lds + (t & 1) * (BJ * STRIDE);
    const bool more = t + 1 < n_tiles;
    if (more) load_tile(j_begin + BJ * (t + 1));

    // S^T tile: c[j][i] = sum_k Z[j0 + j][k] * Z[i0 + i][k]
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    const float* arow = buf + m * STRIDE + h * HALF;
    // The A fragment of the next four k-steps is requested before the current four MFMAs are issued, so that the LDS
    // latency hides behind them.  sched_barrier(0) pins that order: left alone, the scheduler (register pressure is near the
    // 512 budget) sinks every read to just before its use and the wavefront waits ~100 cycles per four MFMAs.
    f32x4 a = *reinterpret_cast<const f32x4*>(arow);
    static_for<HALF / 4>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      f32x4 an = a;
      if (q + 1 < HALF / 4) an = *reinterpret_cast<const f32x4*>(arow + 4 * (q + 1));
      __builtin_amdgcn_sched_barrier(0);
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, zi[4 * q + 0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, zi[4 * q + 1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, zi[4 * q + 2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, zi[4 * q + 3], c, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a = an;
    });

    // softplus for the loss, sigmoid for the gradient; register r of lane half h is row j = 8 (r / 4) + 4 h + r % 4 of the tile
    const int jb = j_begin + BJ * t + 4 * h;
    float tile_loss = 0.f;
    float sg[16];
    auto act = [&](int r) __attribute__((always_inline)) {
      const int j = jb + 8 * (r >> 2) + (r & 3);
      const float x = c[r];
      const float e = __expf(-fabsf(x));
      const float inv = __builtin_amdgcn_rcpf(1.f + e);
      const float sig = x >= 0.f ? inv : e * inv;
      float fval, gval;
      if constexpr (MODE == 0) {
        fval = fmaxf(x, 0.f) + __logf(1.f + e);
        gval = sig;
      } else {
        fval = sig * sig;
        gval = 2.f * fval * (x >= 0.f ? e * inv : inv);  // 1 - sigmoid(x) without cancellation
      }
      const bool valid = j < j_end;
      sg[r] = valid ? gval : 0.f;
      tile_loss += valid ? fval : 0.f;
    };

    // O[i][:] += sum_j sigmoid(S[i][j]) Z[j0 + j][:]: A = sigmoid(c) (row i = lane % 32, k = the j of register r in this lane
    // half).  Step r issues the LDS reads of step r + 1 first, then its ten MFMAs with the sigmoid of step r + 1 in their shadow.
    auto load_b = [&](int r, float (&bv)[NT]) __attribute__((always_inline)) {
      const float* brow = buf + (8 * (r >> 2) + 4 * h + (r & 3)) * STRIDE + m;
      static_for<NT>([&](auto y_c) __attribute__((always_inline)) {
        constexpr int y = decltype(y_c)::value;
        bv[y] = brow[32 * y];
      });
    };
    float bc[NT], bn[NT];
    load_b(0, bc);
    act(0);
    static_for<16>([&](auto r_c) __attribute__((always_inline)) {
      constexpr int r = decltype(r_c)::value;
      if (r + 1 < 16) load_b(r + 1, bn);
      __builtin_amdgcn_sched_barrier(0);
      if (r + 1 < 16) act(r + 1);
      static_for<NT>([&](auto y_c) __attribute__((always_inline)) {
        constexpr int y = decltype(y_c)::value;
        O[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(sg[r], bc[y], O[y], 0, 0, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      if (r + 1 < 16) {
        static_for<NT>([&](auto y_c) __attribute__((always_inline)) {
          constexpr int y = decltype(y_c)::value;
          bc[y] = bn[y];
        });
      }
    });
    lsum += (double)tile_loss;

    if (more) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // partial results of this j range: rows up to n_pad exist in the workspace, so no row guard
  float* op = Opart + ((int64_t)split * n_pad + i0) * DP;
  static_for<NT>([&](auto y_c) __attribute__((always_inline)) {
    constexpr int y = decltype(y_c)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) op[(int64_t)(8 * (r >> 2) + 4 * h + (r & 3)) * DP + 32 * y + m] = O[y][r];
  });
  lsum += __shfl_xor(lsum, 32, 64);
  if (h == 0) Lpart[(int64_t)split * n_pad + i0 + m] = lsum;
}

// O[i][c] = sum_s Opart[s][i][c], rowloss[i] = sum_s Lpart[s][i] — fixed order over the splits
__global__ __launch_bounds__(256) void gram_reduce_kernel(int n, int d, int dp, int n_pad, int splits, const float* __restrict__ Opart,
                                                         const double* __restrict__ Lpart, float* __restrict__ O, int64_t ldo,
                                                         float* __restrict__ rowloss) {
  const int64_t total = (int64_t)n * d;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / d), c = (int)(e - (int64_t)i * d);
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += Opart[((int64_t)s * n_pad + i) * dp + c];
    O[(int64_t)i * ldo + c] = acc;
    if (c == 0) {
      double l = 0.0;
      for (int s = 0; s < splits; ++s) l += Lpart[(int64_t)s * n_pad + i];
      rowloss[i] = (float)l;
    }
  }
}

// ---- the y = 1 entries of the target (edges among the batch's own cells), listed as (us[e], vs[e]) ---------------------------
// forward: xe[e] = <z_us, z_vs> and term[e] = p * softplus(-xe) - softplus(xe), the correction that turns the all-zero-target
// loss of the dense pass into the weighted BCE; one wavefront per entry, butterfly reduction (every lane ends with the same sum).
__device__ __forceinline__ float softplus_exact(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void gram_listed_forward_kernel(int64_t n_listed, int d, const float* __restrict__ Z, int64_t ldz,
                                                                 const int32_t* __restrict__ us, const int32_t* __restrict__ vs, float p,
                                                                 float* __restrict__ xe, float* __restrict__ term) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n_listed) return;
  const float* a = Z + (int64_t)us[e] * ldz;
  const float* b = Z + (int64_t)vs[e] * ldz;
  float acc = 0.f;
  for (int c = lane; c < d; c += 64) acc = fmaf(a[c], b[c], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) {
    xe[e] = acc;
    term[e] = p * softplus_exact(-acc) - softplus_exact(acc);
  }
}

// backward: dZ[i, :] = 2 s O[i, :] + s sum_e c_e ([us[e] == i] z[vs[e], :] + [vs[e] == i] z[us[e], :]),  s = scale[0],
// c_e = p (sigmoid(xe) - 1) - sigmoid(xe).  One workgroup per row i, one column per thread.  The entry list is walked in segments
// of 5 x 256 entries: every wavefront scans ITS 256 entries of the segment (four coalesced loads in flight), ballots the entries
// that touch row i into its own LDS list in lane order, and after a barrier all threads apply the five lists in wave order — list
// order, as before: fixed summation order, no atomics, no sort, bit-identical to the first form.  (In that form every one of the
// five wavefronts scanned the WHOLE list with one load in flight: n x n_listed x 5 pair reads — 2.7 GB of L2 traffic and 0.2 ms per
// call at a batch of 8192 cells, 25 of graph-sc's 250 ms epoch.)
constexpr int GL_WAVES = 5, GL_U = 4, GL_SEG = GL_WAVES * 64 * GL_U;
__global__ __launch_bounds__(320) void gram_listed_backward_kernel(int d, int64_t n_listed, const float* __restrict__ Z, int64_t ldz,
                                                                  const float* __restrict__ O, int64_t ldo, const int32_t* __restrict__ us,
                                                                  const int32_t* __restrict__ vs, const float* __restrict__ xe, float p,
                                                                  const float* __restrict__ scale, float* __restrict__ dZ, int64_t ldd) {
  __shared__ int hu[GL_WAVES][64 * GL_U], hv[GL_WAVES][64 * GL_U];
  __shared__ float hc[GL_WAVES][64 * GL_U];
  __shared__ int hn[GL_WAVES];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = threadIdx.x;  // column owned by this thread (threads beyond d only help scanning)
  const float s = scale[0];
  float acc = c < d ? 2.f * O[(int64_t)i * ldo + c] : 0.f;
  for (int64_t seg = 0; seg < n_listed; seg += GL_SEG) {
    int u[GL_U], v[GL_U];
#pragma unroll
    for (int q = 0; q < GL_U; ++q) {  // clamped addresses: the loads are not behind a branch, all four are in flight
      const int64_t e = seg + (int64_t)wave * (64 * GL_U) + q * 64 + lane;
      const int64_t ec = e < n_listed ? e : n_listed - 1;
      u[q] = us[ec];
      v[q] = vs[ec];
      if (e >= n_listed) u[q] = v[q] = -1;
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < GL_U; ++q) {
      const bool hit = (u[q] == i) || (v[q] == i);
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        const float xx = xe[seg + (int64_t)wave * (64 * GL_U) + q * 64 + lane];
        const float sg = 1.f / (1.f + expf(-xx));
        hu[wave][pos] = u[q];
        hv[wave][pos] = v[q];
        hc[wave][pos] = p * (sg - 1.f) - sg;
      }
      cnt += __popcll(m);
    }
    if (lane == 0) hn[wave] = cnt;
    __syncthreads();
    if (c < d) {
      for (int w = 0; w < GL_WAVES; ++w) {
        const int nh = hn[w];
        for (int k = 0; k < nh; ++k) {
          const int uu = hu[w][k], vv = hv[w][k];
          const float ce = hc[w][k];
          if (uu == i) acc = fmaf(ce, Z[(int64_t)vv * ldz + c], acc);
          if (vv == i) acc = fmaf(ce, Z[(int64_t)uu * ldz + c], acc);
        }
      }
    }
    __syncthreads();
  }
  if (c < d) dZ[(int64_t)i * ldd + c] = s * acc;
}

// the listed target is the IDENTITY (every cell's self loop and nothing else: the decoder target of a batch of cells of a cell - gene graph,
// graphsc.py:208-214): dZ[i, :] = 2 s (O[i, :] + c_i z[i, :]) — an elementwise pass instead of n workgroups each scanning the n-entry list
__global__ __launch_bounds__(256) void gram_diag_backward_kernel(int64_t n, int d, const float* __restrict__ Z, int64_t ldz, const float* __restrict__ O,
                                                                int64_t ldo, const float* __restrict__ xe, float p, const float* __restrict__ scale,
                                                                float* __restrict__ dZ, int64_t ldd) {
  const float s = scale[0];
  const int64_t total = n * d;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / d;
    const int c = (int)(e - i * d);
    const float sg = 1.f / (1.f + expf(-xe[i]));
    const float ce = p * (sg - 1.f) - sg;
    // the generic kernel's order: acc = 2 O; acc = fma(ce, z, acc) twice (u == i and v == i); s * acc
    float acc = 2.f * O[i * ldo + c];
    const float z = Z[i * ldz + c];
    acc = fmaf(ce, z, acc);
    acc = fmaf(ce, z, acc);
    dZ[i * ldd + c] = s * acc;
  }
}

struct Plan {
  int dp, i_blocks, splits, j_per_split, n_pad;
  size_t opart_bytes, lpart_bytes;
};

Plan make_plan(int64_t n_r, int64_t n, int64_t d) {  // n_r rows i against n columns j
  Plan p{};
  p.dp = (int)(dh::ceil_div(d, 64) * 64);
  p.i_blocks = (int)dh::ceil_div(n_r, BI);
  p.n_pad = p.i_blocks * BI;
  const int64_t j_tiles = dh::ceil_div(n, BJ);
  // one workgroup per CU (the kernel runs ONE wave per SIMD: two never share a CU); until round 6 this asked for 512 workgroups, i.e. two
  // rounds of half-length j ranges with twice the prologues, partial outputs and reduce traffic: 0.822 -> 0.80 ms at 8192 x 300
  int64_t want = dh::ceil_div(256, p.i_blocks);
  if (want > j_tiles) want = j_tiles;
  if (want < 1) want = 1;
  p.j_per_split = (int)(dh::ceil_div(j_tiles, want) * BJ);
  p.splits = (int)dh::ceil_div(n, p.j_per_split);
  p.opart_bytes = (size_t)p.splits * p.n_pad * p.dp * sizeof(float);
  p.lpart_bytes = (size_t)p.splits * p.n_pad * sizeof(double);
  return p;
}

template <int DP, int MODE>
int launch(const Plan& p, int n_r, const float* Zr, int64_t ldr, int n, int d, const float* Z, int64_t ldz, float* Opart, double* Lpart, hipStream_t st) {
  const bool vec = (ldz % 4 == 0) && (ldr % 4 == 0) && (d % 4 == 0) && dh::aligned16(Z) && dh::aligned16(Zr);
  const size_t lds_bytes = (size_t)2 * BJ * lds_stride(DP) * sizeof(float);
  const dim3 grid((unsigned)p.i_blocks, (unsigned)p.splits);
  if (vec) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gram_sigmoid_kernel<DP, true, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((gram_sigmoid_kernel<DP, true, MODE>), grid, dim3(256), lds_bytes, st, n_r, Zr, ldr, n, d, Z, ldz, p.j_per_split, p.n_pad, Opart, Lpart);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gram_sigmoid_kernel<DP, false, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((gram_sigmoid_kernel<DP, false, MODE>), grid, dim3(256), lds_bytes, st, n_r, Zr, ldr, n, d, Z, ldz, p.j_per_split, p.n_pad, Opart, Lpart);
  }
  return dh::check_launch("dh_gram_pairwise_f32");
}

template <int MODE>
int launch_dp(const Plan& p, int n_r, const float* Zr, int64_t ldr, int n, int d, const float* Z, int64_t ldz, float* Opart, double* Lpart, hipStream_t st) {
  switch (p.dp) {
    case 64: return launch<64, MODE>(p, n_r, Zr, ldr, n, d, Z, ldz, Opart, Lpart, st);
    case 128: return launch<128, MODE>(p, n_r, Zr, ldr, n, d, Z, ldz, Opart, Lpart, st);
    case 192: return launch<192, MODE>(p, n_r, Zr, ldr, n, d, Z, ldz, Opart, Lpart, st);
    case 256: return launch<256, MODE>(p, n_r, Zr, ldr, n, d, Z, ldz, Opart, Lpart, st);
    default: return launch<320, MODE>(p, n_r, Zr, ldr, n, d, Z, ldz, Opart, Lpart, st);
  }
}

}  // namespace

extern "C" int dh_gram_sigmoid_supported(int64_t n, int64_t d) { return n >= 0 && n <= (int64_t)1 << 24 && d >= 1 && d <= MAX_D; }

extern "C" size_t dh_gram_sigmoid_workspace_bytes(int64_t n, int64_t d) {
  if (n <= 0 || !dh_gram_sigmoid_supported(n, d)) return 0;
  const Plan p = make_plan(n, n, d);
  return p.opart_bytes + p.lpart_bytes;
}

extern "C" size_t dh_gram_pairwise_rect_workspace_bytes(int64_t n_rows, int64_t n, int64_t d) {
  if (n_rows <= 0 || n <= 0 || !dh_gram_sigmoid_supported(n, d) || !dh_gram_sigmoid_supported(n_rows, d)) return 0;
  const Plan p = make_plan(n_rows, n, d);
  return p.opart_bytes + p.lpart_bytes;
}

extern "C" int dh_gram_sigmoid_f32(int64_t n, int64_t d, const float* Z, int64_t ldz, float* O, int64_t ldo, float* rowloss,
                                   void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  return dh_gram_pairwise_f32(DH_GRAM_SOFTPLUS, n, d, Z, ldz, O, ldo, rowloss, workspace, workspace_bytes, stream);
}

extern "C" int dh_gram_pairwise_f32(int mode, int64_t n, int64_t d, const float* Z, int64_t ldz, float* O, int64_t ldo, float* rowloss,
                                    void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  return dh_gram_pairwise_rect_f32(mode, n, n, d, Z, ldz, Z, ldz, O, ldo, rowloss, workspace, workspace_bytes, stream);
}

extern "C" int dh_gram_pairwise_rect_f32(int mode, int64_t n_rows, int64_t n, int64_t d, const float* Zr, int64_t ldr, const float* Z,
                                         int64_t ldz, float* O, int64_t ldo, float* rowloss, void* workspace, size_t workspace_bytes,
                                         dh_stream_t stream) {
  const char* me = "dh_gram_pairwise_rect_f32";
  if (mode != DH_GRAM_SOFTPLUS && mode != DH_GRAM_SIGMOID_SQ) return dh::fail(DH_ERR_INVALID, "%s: bad mode %d", me, mode);
  if (n_rows < 0 || n < 0 || d < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_rows == 0) return DH_OK;
  if (!dh_gram_sigmoid_supported(n, d) || !dh_gram_sigmoid_supported(n_rows, d))
    return dh::fail(DH_ERR_INVALID, "%s: d = %lld outside [1, %d] (use the unfused decoder)", me, (long long)d, MAX_D);
  if (!Zr || (n > 0 && !Z) || !O || !rowloss || ldz < d || ldr < d || ldo < d) return dh::fail(DH_ERR_INVALID, "%s: bad pointer / leading dimension", me);
  hipStream_t st = dh::as_stream(stream);
  if (n == 0) {  // no columns: empty sums
    if (dh::zero2d_async(O, (size_t)ldo * sizeof(float), (size_t)d * sizeof(float), (size_t)n_rows, st) != hipSuccess ||
        dh::zero_async(rowloss, (size_t)n_rows * sizeof(float), st) != hipSuccess)
      return dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me);
    return DH_OK;
  }
  const Plan p = make_plan(n_rows, n, d);
  if (!workspace || workspace_bytes < p.opart_bytes + p.lpart_bytes)
    return dh::fail(DH_ERR_INVALID, "%s: workspace of %zu bytes needed, %zu given", me, p.opart_bytes + p.lpart_bytes, workspace_bytes);
  if (!dh::aligned16(workspace)) return dh::fail(DH_ERR_INVALID, "%s: workspace must be 16-byte aligned", me);
  double* Lpart = reinterpret_cast<double*>(workspace);                      // doubles first: keeps both parts aligned
  float* Opart = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + p.lpart_bytes);
  const int rc = mode == DH_GRAM_SOFTPLUS ? launch_dp<0>(p, (int)n_rows, Zr, ldr, (int)n, (int)d, Z, ldz, Opart, Lpart, st)
                                          : launch_dp<1>(p, (int)n_rows, Zr, ldr, (int)n, (int)d, Z, ldz, Opart, Lpart, st);
  if (rc != DH_OK) return rc;
  const int64_t work = n_rows * d;
  const unsigned grid = (unsigned)(dh::ceil_div(work, 256) < 65536 ? dh::ceil_div(work, 256) : 65536);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(grid), dim3(256), 0, st, (int)n_rows, (int)d, p.dp, p.n_pad, p.splits, Opart, Lpart, O, ldo, rowloss);
  return dh::check_launch("dh_gram_pairwise_rect_f32 (reduce)");
}

extern "C" int dh_gram_listed_forward_f32(int64_t n, int64_t d, int64_t n_listed, const float* Z, int64_t ldz, const int32_t* us,
                                          const int32_t* vs, float pos_weight, float* xe, float* term, dh_stream_t stream) {
  if (n < 0 || d < 0 || n_listed < 0) return dh::fail(DH_ERR_INVALID, "dh_gram_listed_forward_f32: negative size");
  if (n_listed == 0) return DH_OK;
  if (!Z || !us || !vs || !xe || !term || ldz < d) return dh::fail(DH_ERR_INVALID, "dh_gram_listed_forward_f32: bad pointer / leading dimension");
  hipLaunchKernelGGL(gram_listed_forward_kernel, dim3((unsigned)dh::ceil_div(n_listed, 4)), dim3(256), 0, dh::as_stream(stream), n_listed, (int)d, Z,
                     ldz, us, vs, pos_weight, xe, term);
  return dh::check_launch("dh_gram_listed_forward_f32");
}

extern "C" int dh_gram_diag_backward_f32(int64_t n, int64_t d, const float* Z, int64_t ldz, const float* O, int64_t ldo, const float* xe, float pos_weight,
                                         const float* scale, float* dZ, int64_t ldd, dh_stream_t stream) {
  if (n < 0 || d < 0) return dh::fail(DH_ERR_INVALID, "dh_gram_diag_backward_f32: negative size");
  if (n == 0 || d == 0) return DH_OK;
  if (!Z || !O || !xe || !scale || !dZ || ldz < d || ldo < d || ldd < d) return dh::fail(DH_ERR_INVALID, "dh_gram_diag_backward_f32: bad pointer / leading dimension");
  const unsigned grid = (unsigned)std::min<int64_t>(dh::ceil_div(n * d, 256), 8192);
  hipLaunchKernelGGL(gram_diag_backward_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), n, (int)d, Z, ldz, O, ldo, xe, pos_weight, scale, dZ, ldd);
  return dh::check_launch("dh_gram_diag_backward_f32");
}

extern "C" int dh_gram_listed_backward_f32(int64_t n, int64_t d, int64_t n_listed, const float* Z, int64_t ldz, const float* O, int64_t ldo,
                                           const int32_t* us, const int32_t* vs, const float* xe, float pos_weight, const float* scale,
                                           float* dZ, int64_t ldd, dh_stream_t stream) {
  if (n < 0 || d < 0 || n_listed < 0) return dh::fail(DH_ERR_INVALID, "dh_gram_listed_backward_f32: negative size");
  if (n == 0 || d == 0) return DH_OK;
  if (d > MAX_D) return dh::fail(DH_ERR_INVALID, "dh_gram_listed_backward_f32: d = %lld outside [1, %d]", (long long)d, MAX_D);
  if (!Z || !O || !scale || !dZ || ldz < d || ldo < d || ldd < d || (n_listed > 0 && (!us || !vs || !xe)))
    return dh::fail(DH_ERR_INVALID, "dh_gram_listed_backward_f32: bad pointer / leading dimension");
  hipLaunchKernelGGL(gram_listed_backward_kernel, dim3((unsigned)n), dim3(320), 0, dh::as_stream(stream), (int)d, n_listed, Z, ldz, O, ldo, us, vs, xe,
                     pos_weight, scale, dZ, ldd);
  return dh::check_launch("dh_gram_listed_backward_f32");
}
