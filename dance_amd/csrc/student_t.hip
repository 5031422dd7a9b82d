// Student-t soft assignment of the DEC clustering heads (SpaGCN's SimpleGCDEC / GC_DEC, spagcn.py:391-397,600-608; scDSC,
// scdsc.py:466-468) as ONE kernel each way:
//     d2_ij = ||z_i - mu_j||^2,   base_ij = 1 / ((1 + d2_ij / a) + eps),   u_ij = base_ij^pw * scale,   q_ij = u_ij / sum_j u_ij
// The reference writes it with broadcasting — z.unsqueeze(1) - mu is an [N, C, d] tensor (1 GB at 500k spots x 10 x 50, 1.3 GB at 1M
// cells x 10 x 32), written, squared, summed, and walked again by autograd: at BASELINE config 5 the head cost 4.1 of an iteration's
// 4.8 ms, six times the layer's own two kernels (profiles/r05e_configs.json).  Here z is read once per pass (N d floats), q written
// once (N C floats), nothing of size N C d exists.  HBM-bound by N (d + C) 4 bytes forward, N (2 d + 2 C) 4 backward.
//
// Mapping: one thread per row, 128 rows per block.  mu (C x d <= 4096 floats) sits in LDS and is read as a broadcast (every lane the
// same address); the block's rows of z are brought into LDS with coalesced loads and walked per lane (odd row stride: no bank
// conflicts) ONCE per group of 16 clusters, whose squared distances accumulate in registers.  The backward recomputes base / u / S
// from z and mu instead of storing them, forms
//     c_ij = dL / d d2_ij = -(G_ij - sum_k G_ik q_ik) q_ij pw base_ij / a,
//     dz_i = 2 (z_i sum_j c_ij - sum_j c_ij mu_j),     dmu_j = -2 (sum_i c_ij z_i - mu_j sum_i c_ij)
// keeps the row's c_ij in LDS, and reduces dmu over the block's rows with the threads re-dealt over (j, t) pairs (coalesced reads of
// the rows the block just touched); block partials are summed by a second kernel in block order — no float atomics, deterministic.
#include "common.h"

namespace {

constexpr int ST_ROWS = 128;   // rows (= threads) per block
constexpr int ST_MAX_C = 64;
constexpr int ST_MAX_CD = 4096;
// LDS of the backward kernel: mu + the rows' coefficients + the rows of z, within the 64 KB a launch gets without opting in
__host__ __device__ constexpr int64_t st_lds_floats(int64_t c, int64_t d) { return c * d + ST_ROWS * (c | 1) + ST_ROWS * (d | 1); }
constexpr int64_t ST_MAX_LDS_FLOATS = 16384;

struct StParams {
  float a, eps, pw, scale;
};

__device__ __forceinline__ float st_base(float d2, const StParams& p) { return 1.f / ((1.f + d2 / p.a) + p.eps); }

constexpr int ST_CG = 16;  // clusters whose squared distances a lane accumulates in registers during one walk over its row

// d2[jj] = ||z_row - mu_(g0 + jj)||^2 for jj < ST_CG: ONE walk over the row (the first version walked it once per cluster: C d loads
// per row per pass made the kernels texture-path-bound: 0.71 / 1.70 ms forward / backward at 500k x 10 x 50 instead of ~0.1).
// zr: the lane's row, in global memory (stride 1) or in the block's LDS tile; clusters past c repeat the last one (ignored by the caller).
__device__ __forceinline__ void st_dist_group(float (&d2)[ST_CG], const float* __restrict__ zr, const float* __restrict__ mu, int g0, int c, int d) {
#pragma unroll
  for (int jj = 0; jj < ST_CG; ++jj) d2[jj] = 0.f;
  for (int t = 0; t < d; ++t) {
    const float z = zr[t];
#pragma unroll
    for (int jj = 0; jj < ST_CG; ++jj) {
      const int j = min(g0 + jj, c - 1);
      const float diff = z - mu[j * d + t];
      d2[jj] = fmaf(diff, diff, d2[jj]);
    }
  }
}

__global__ __launch_bounds__(ST_ROWS) void student_t_forward_kernel(int64_t n, int c, int d, const float* __restrict__ Z, int64_t ldz,
                                                                    const float* __restrict__ MU, StParams p, float* __restrict__ Q, int64_t ldq) {
  extern __shared__ float smem[];  // mu [c][d], then the block's rows of z [ST_ROWS][d | 1] (coalesced in, walked per lane without bank conflicts)
  float* mu = smem;
  const int zs = d | 1;
  float* zt = smem + c * d;
  for (int i = threadIdx.x; i < c * d; i += ST_ROWS) mu[i] = MU[i];
  const int64_t row0 = (int64_t)blockIdx.x * ST_ROWS;
  const int rows_here = (int)min((int64_t)ST_ROWS, n - row0);
  for (int i = threadIdx.x; i < rows_here * d; i += ST_ROWS) {
    const int r = i / d, t = i - r * d;
    zt[r * zs + t] = Z[(row0 + r) * ldz + t];
  }
  __syncthreads();
  if ((int)threadIdx.x >= rows_here) return;
  const float* zr = zt + threadIdx.x * zs;
  float* qr = Q + (row0 + threadIdx.x) * ldq;
  float s = 0.f;
  for (int g0 = 0; g0 < c; g0 += ST_CG) {
    float d2[ST_CG];
    st_dist_group(d2, zr, mu, g0, c, d);
#pragma unroll
    for (int jj = 0; jj < ST_CG; ++jj)
      if (g0 + jj < c) {
        const float u = powf(st_base(d2[jj], p), p.pw) * p.scale;
        qr[g0 + jj] = u;
        s += u;
      }
  }
  for (int j = 0; j < c; ++j) qr[j] = qr[j] / s;  // the row just written by this lane: L1 / L2 resident
}

// dZ (may be null), and this block's partial of dMU: part[block][c * d + c] = [sum_i c_ij z_it | sum_i c_ij]
__global__ __launch_bounds__(ST_ROWS) void student_t_backward_kernel(int64_t n, int c, int d, const float* __restrict__ Z, int64_t ldz,
                                                                     const float* __restrict__ MU, StParams p, const float* __restrict__ G,
                                                                     int64_t ldg, float* __restrict__ dZ, int64_t lddz, float* __restrict__ part) {
  extern __shared__ float smem[];
  const int cs_ = c | 1, zs = d | 1;   // odd row strides: a lane walking its own row hits its own bank
  float* mu = smem;                    // [c][d]
  float* cf = smem + c * d;            // [ST_ROWS][c | 1]  the rows' squared distances, then their c_ij (0 for rows past n)
  float* zt = cf + ST_ROWS * cs_;      // [ST_ROWS][d | 1]  the block's rows of z
  for (int i = threadIdx.x; i < c * d; i += ST_ROWS) mu[i] = MU[i];
  const int64_t row0 = (int64_t)blockIdx.x * ST_ROWS;
  const int rows_here = (int)min((int64_t)ST_ROWS, n - row0);
  for (int i = threadIdx.x; i < rows_here * d; i += ST_ROWS) {
    const int r = i / d, t = i - r * d;
    zt[r * zs + t] = Z[(row0 + r) * ldz + t];
  }
  __syncthreads();
  float* my_c = cf + threadIdx.x * cs_;
  if ((int)threadIdx.x < rows_here) {
    const int64_t row = row0 + threadIdx.x;
    const float* zr = zt + threadIdx.x * zs;
    const float* gr = G + row * ldg;
    float s = 0.f, t_gq = 0.f;
    for (int g0 = 0; g0 < c; g0 += ST_CG) {   // pass 1: distances (kept in LDS), S, sum_j G_j u_j
      float d2[ST_CG];
      st_dist_group(d2, zr, mu, g0, c, d);
#pragma unroll
      for (int jj = 0; jj < ST_CG; ++jj)
        if (g0 + jj < c) {
          const float u = powf(st_base(d2[jj], p), p.pw) * p.scale;
          my_c[g0 + jj] = d2[jj];
          s += u;
          t_gq = fmaf(gr[g0 + jj], u, t_gq);
        }
    }
    const float rs = 1.f / s;
    t_gq *= rs;                        // sum_k G_k q_k
    float csum = 0.f;
    for (int j = 0; j < c; ++j) {      // pass 2: c_j from the stored distance
      const float base = st_base(my_c[j], p);
      const float q = powf(base, p.pw) * p.scale * rs;
      const float cj = -(gr[j] - t_gq) * q * p.pw * base / p.a;
      my_c[j] = cj;
      csum += cj;
    }
    if (dZ) {
      float* dzr = dZ + row * lddz;
      for (int t = 0; t < d; ++t) {
        float acc = zr[t] * csum;
        for (int j = 0; j < c; ++j) acc = fmaf(-my_c[j], mu[j * d + t], acc);
        dzr[t] = 2.f * acc;
      }
    }
  } else {
    for (int j = 0; j < c; ++j) my_c[j] = 0.f;
  }
  __syncthreads();
  // the block's share of dMU: pairs (j, t) dealt over the threads, rows in order (deterministic); column d of a cluster = sum_i c_ij
  float* out = part + (int64_t)blockIdx.x * (c * d + c);
  for (int pair = threadIdx.x; pair < c * d; pair += ST_ROWS) {
    const int j = pair / d, t = pair - j * d;
    float acc = 0.f;
    for (int i = 0; i < rows_here; ++i) acc = fmaf(cf[i * cs_ + j], zt[i * zs + t], acc);
    out[pair] = acc;
  }
  for (int j = threadIdx.x; j < c; j += ST_ROWS) {
    float acc = 0.f;
    for (int i = 0; i < rows_here; ++i) acc += cf[i * cs_ + j];
    out[c * d + j] = acc;
  }
}

// dMU[j][t] = -2 (sum_blocks part[b][j d + t] - mu[j][t] sum_blocks part[b][c d + j]): one workgroup per (j, t); thread i sums the
// blocks i, i + 256, ... in order, then a fixed tree over the 256 partial sums (deterministic)
__global__ __launch_bounds__(256) void student_t_reduce_kernel(int64_t n_blocks, int c, int d, const float* __restrict__ part,
                                                               const float* __restrict__ MU, float* __restrict__ dMU) {
  __shared__ float red[2][256];
  const int pair = blockIdx.x;
  const int j = pair / d;
  const int64_t stride = (int64_t)c * d + c;
  float cz = 0.f, cs = 0.f;
  for (int64_t b = threadIdx.x; b < n_blocks; b += 256) {
    cz += part[b * stride + pair];
    cs += part[b * stride + (int64_t)c * d + j];
  }
  red[0][threadIdx.x] = cz;
  red[1][threadIdx.x] = cs;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) dMU[pair] = -2.f * (red[0][0] - MU[pair] * red[1][0]);
}

// ---- round 5, second form: the same two passes for c <= 32 clusters with a fifth of the LDS instructions -------------------------------
// The kernels above issue one LDS read per (row, cluster, feature) for mu and one per (row, feature) for z, the backward another 2 per
// (pair, row) for the dMU partial: ~850 + 1000 + 1000 wave-level LDS instructions per 64 rows at 10 x 50, which is what bounded them
// (0.155 / 0.30 ms at 500k rows; the bytes are worth 0.015 / 0.03).  Here
//   * mu is kept TRANSPOSED and zero padded, muT[t][CP] (CP = c rounded up to 4): the CP clusters of one feature are CP / 4 broadcast
//     ds_read_b128; a lane reads its own row of z as b128 too (row stride 4 x odd: the 16 lanes of a quarter wave hit 64 distinct banks);
//   * the block's share of dMU = C^T Z ([c x 128] x [128 x d], plus the column sums of C as the product with a column of ones) runs on
//     the matrix cores (v_mfma_f32_16x16x4_f32, both operands straight from the LDS tiles);
//   * Q, G and dZ move between HBM and LDS as whole tiles (coalesced) instead of one 4-byte store per lane and cluster / feature;
//   * base^pw is exp2(pw log2 base) on the transcendental unit (relative error ~1e-6 at base = 1e-3; exact for pw = 1) instead of powf;
//   * no load of the tile loops sits behind a branch (clamped address + select).
constexpr int STF_ROWS = 128;
typedef float stf_f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int stf_d4(int d) { return (d + 3) & ~3; }
__host__ __device__ constexpr int stf_zs(int d) { return 4 * (((d + 3) / 4) | 1); }  // z tile row stride: a multiple of 4 floats, stride / 4 odd
__host__ __device__ constexpr int stf_cp(int c) { return (c + 3) & ~3; }
__host__ __device__ constexpr int64_t stf_lds_floats(int c, int d) {
  return (int64_t)stf_d4(d) * stf_cp(c) + (int64_t)STF_ROWS * stf_zs(d) + (int64_t)STF_ROWS * (stf_cp(c) + 1);
}

__device__ __forceinline__ float stf_pow(float base, float pw) { return pw == 1.f ? base : __expf(pw * __logf(base)); }

template <int VW>
__device__ __forceinline__ void stf_stage_z(float* zt, int64_t n, int d, const float* __restrict__ Z, int64_t ldz, int64_t row0) {
  typedef float VT __attribute__((ext_vector_type(VW)));
  constexpr int LPR = 64 / VW;  // lanes per row
  const int d4 = stf_d4(d), zs = stf_zs(d);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane / LPR, cl = (lane % LPR) * VW;
  const int dv = (d + VW - 1) / VW * VW;  // the last vector of a row may reach past d: inside the row's stride (ldz % VW == 0, ldz >= d)
  for (int t0 = 0; t0 < d4; t0 += 64) {
    const int t = t0 + cl;
    if (t < d4) {
      const int tc = min(t, dv - VW);
      constexpr int NIT = STF_ROWS / (2 * VW);  // 16 / 32 / 64 row groups per wave
#pragma unroll 1
      for (int k0 = 0; k0 < NIT; k0 += 16) {
        VT v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int r = ((k0 + u) * 2 + wv) * VW + sub;
          v[u] = *reinterpret_cast<const VT*>(Z + min(row0 + r, n - 1) * ldz + tc);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int r = ((k0 + u) * 2 + wv) * VW + sub;
          VT o;
#pragma unroll
          for (int i = 0; i < VW; ++i) o[i] = (row0 + r < n && t + i < d) ? v[u][i] : 0.f;
          *reinterpret_cast<VT*>(zt + r * zs + t) = o;
        }
      }
    }
  }
}

// muT and the z tile (rows past n and columns past d are zeros: such a row has d2 = |mu|^2, finite, and is never stored)
template <int CP>
__device__ __forceinline__ void stf_stage(float* muT, float* zt, int64_t n, int c, int d, const float* __restrict__ Z, int64_t ldz,
                                          const float* __restrict__ MU, int64_t row0) {
  const int d4 = stf_d4(d);
  for (int i = threadIdx.x; i < d4 * CP; i += STF_ROWS) {
    const int t = i / CP, j = i - t * CP;
    const float v = MU[min(j, c - 1) * d + min(t, d - 1)];
    muT[i] = (t < d && j < c) ? v : 0.f;
  }
  // A pass covers 64 columns: 64 / VW lanes per row with VW floats each, VW rows per wave instruction, the two waves on alternate row
  // groups; all 16 loads of a lane are issued before the first is used (no division, no branch between them).  At VW = 1 (the first
  // version) a wave needed 8 dependent round trips per tile, and with 8 waves per CU (LDS) that latency, not the bytes, was the kernel.
  const bool v4 = ldz % 4 == 0 && (reinterpret_cast<uintptr_t>(Z) & 15) == 0, v2 = ldz % 2 == 0 && (reinterpret_cast<uintptr_t>(Z) & 7) == 0;
  if (v4) stf_stage_z<4>(zt, n, d, Z, ldz, row0);
  else if (v2) stf_stage_z<2>(zt, n, d, Z, ldz, row0);
  else stf_stage_z<1>(zt, n, d, Z, ldz, row0);
}

// rows [0, rows_here) x columns [0, d) of an LDS tile -> global memory, the same lane-per-column walk
__device__ __forceinline__ void stf_store_rows(const float* tile, int ts, int rows_here, int d, float* __restrict__ out, int64_t ld, int64_t row0) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int t = lane; t < d; t += 64)
    for (int r = wv; r < rows_here; r += STF_ROWS / 64) out[(row0 + r) * ld + t] = tile[r * ts + t];
}

__device__ __forceinline__ float stf_base(float d2, float inv_a, float eps) { return __frcp_rn((1.f + d2 * inv_a) + eps); }

// d2[j] = ||z_row - mu_j||^2, j < CP (padded clusters: |z|^2, ignored)
template <int CP>
__device__ __forceinline__ void stf_dist(float (&d2)[CP], const float* zr, const float* muT, int d4) {
#pragma unroll
  for (int j = 0; j < CP; ++j) d2[j] = 0.f;
  for (int t4 = 0; t4 < d4; t4 += 4) {
    const stf_f32x4 z4 = *reinterpret_cast<const stf_f32x4*>(zr + t4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < CP / 4; ++q) {
        const stf_f32x4 m4 = *reinterpret_cast<const stf_f32x4*>(muT + (t4 + e) * CP + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float diff = z4[e] - m4[i];
          d2[4 * q + i] = fmaf(diff, diff, d2[4 * q + i]);
        }
      }
  }
}

template <int CP>
__global__ __launch_bounds__(STF_ROWS) void student_t_forward_fast_kernel(int64_t n, int c, int d, const float* __restrict__ Z, int64_t ldz,
                                                                          const float* __restrict__ MU, StParams p, float* __restrict__ Q, int64_t ldq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d4 = stf_d4(d), zs = stf_zs(d);
  constexpr int cs = CP + 1;
  float* muT = smem;
  float* zt = muT + d4 * CP;
  float* qt = zt + STF_ROWS * zs;
  const int64_t row0 = (int64_t)blockIdx.x * STF_ROWS;
  const int rows_here = (int)min((int64_t)STF_ROWS, n - row0);
  stf_stage<CP>(muT, zt, n, c, d, Z, ldz, MU, row0);
  __syncthreads();
  float d2[CP];
  stf_dist<CP>(d2, zt + threadIdx.x * zs, muT, d4);
  const float inv_a = 1.f / p.a;
  float u[CP], s = 0.f;
#pragma unroll
  for (int j = 0; j < CP; ++j) {
    u[j] = j < c ? stf_pow(stf_base(d2[j], inv_a, p.eps), p.pw) * p.scale : 0.f;
    s += u[j];
  }
  const float rs = __frcp_rn(s);
#pragma unroll
  for (int j = 0; j < CP; ++j) qt[threadIdx.x * cs + j] = u[j] * rs;
  __syncthreads();
  // the padded index space [128][CP]: consecutive lanes = consecutive clusters of consecutive rows (contiguous in Q when ldq = c)
  for (int i = threadIdx.x; i < rows_here * CP; i += STF_ROWS) {
    const int r = i / CP, j = i - r * CP;
    if (j < c) Q[(row0 + r) * ldq + j] = qt[r * cs + j];
  }
}

template <int CP>
__global__ __launch_bounds__(STF_ROWS) void student_t_backward_fast_kernel(int64_t n, int c, int d, const float* __restrict__ Z, int64_t ldz,
                                                                           const float* __restrict__ MU, StParams p, const float* __restrict__ G,
                                                                           int64_t ldg, float* __restrict__ dZ, int64_t lddz, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d4 = stf_d4(d), zs = stf_zs(d);
  constexpr int cs = CP + 1;
  float* muT = smem;
  float* zt = muT + d4 * CP;
  float* gt = zt + STF_ROWS * zs;  // the rows' upstream gradients G, then their coefficients c_ij
  const int64_t row0 = (int64_t)blockIdx.x * STF_ROWS;
  const int rows_here = (int)min((int64_t)STF_ROWS, n - row0);
  stf_stage<CP>(muT, zt, n, c, d, Z, ldz, MU, row0);
  {
    float gv[CP];  // CP loads per lane, ALL issued before the first LDS store (written as one loop the compiler waited after every load)
#pragma unroll
    for (int k = 0; k < CP; ++k) {
      const int i = threadIdx.x + STF_ROWS * k;
      const int r = i / CP, j = i - r * CP;
      gv[k] = G[min(row0 + r, n - 1) * ldg + min(j, c - 1)];
    }
#pragma unroll
    for (int k = 0; k < CP; ++k) {
      const int i = threadIdx.x + STF_ROWS * k;
      const int r = i / CP, j = i - r * CP;
      gt[r * cs + j] = (row0 + r < n && j < c) ? gv[k] : 0.f;
    }
  }
  __syncthreads();
  // per row: the coefficients c_j (registers, and the row's slot of gt) — a row past n has G = 0, hence c_j = 0: no special case
  float cj[CP], csum = 0.f;
  {
    float d2[CP];
    stf_dist<CP>(d2, zt + threadIdx.x * zs, muT, d4);
    const float inv_a = 1.f / p.a;
    float base[CP], s = 0.f, t_gq = 0.f;
#pragma unroll
    for (int j = 0; j < CP; ++j) {
      base[j] = stf_base(d2[j], inv_a, p.eps);
      cj[j] = j < c ? stf_pow(base[j], p.pw) * p.scale : 0.f;  // u_j
      s += cj[j];
      t_gq = fmaf(gt[threadIdx.x * cs + j], cj[j], t_gq);
    }
    const float rs = __frcp_rn(s);
    t_gq *= rs;  // sum_k G_k q_k
    const float k_pa = p.pw * inv_a;
#pragma unroll
    for (int j = 0; j < CP; ++j) {
      const float q = cj[j] * rs;
      cj[j] = -(gt[threadIdx.x * cs + j] - t_gq) * q * k_pa * base[j];
      csum += cj[j];
    }
#pragma unroll
    for (int j = 0; j < CP; ++j) gt[threadIdx.x * cs + j] = cj[j];
  }
  __syncthreads();
  // the block's share of dMU on the matrix cores: P[i][f] = sum_r c_ri z_rf (f < d), P[i][d] = sum_r c_ri.  16 x 16 tiles, K = 4 rows
  // per step: lane l holds A[i = l % 16][k = l / 16] = c of row 4 s + l / 16, B[k][f = l % 16] = z of the same row; D[4 (l / 16) + r][l % 16]
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int n_tiles = (d + 1 + 15) / 16;
    float* out = part + (int64_t)blockIdx.x * (c * d + c);
    for (int nt = wave; nt < n_tiles; nt += STF_ROWS / 64) {
      const int f = 16 * nt + l16;
#pragma unroll
      for (int mt = 0; mt < (CP + 15) / 16; ++mt) {
        const int ci = 16 * mt + l16;
        stf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < STF_ROWS / 4; ++ks) {
          const int r = 4 * ks + kq;
          const float a = ci < CP ? gt[r * cs + min(ci, CP - 1)] : 0.f;
          const float zv = zt[r * zs + min(f, d4 - 1)];
          const float b = f < d ? zv : (f == d ? 1.f : 0.f);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * mt + 4 * kq + r;
          if (i < c) {
            if (f < d) out[i * d + f] = acc[r];
            else if (f == d) out[c * d + i] = acc[r];
          }
        }
      }
    }
  }
  if (!dZ) return;  // uniform
  __syncthreads();  // the product above read every row of zt; below each lane replaces its own row by dz
  {
    float* zr = zt + threadIdx.x * zs;
    for (int t4 = 0; t4 < d4; t4 += 4) {
      const stf_f32x4 z4 = *reinterpret_cast<const stf_f32x4*>(zr + t4);
      stf_f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = z4[e] * csum;
#pragma unroll
        for (int q = 0; q < CP / 4; ++q) {
          const stf_f32x4 m4 = *reinterpret_cast<const stf_f32x4*>(muT + (t4 + e) * CP + 4 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = fmaf(-cj[4 * q + i], m4[i], acc);
        }
        o[e] = 2.f * acc;
      }
      *reinterpret_cast<stf_f32x4*>(zr + t4) = o;
    }
  }
  __syncthreads();
  stf_store_rows(zt, zs, rows_here, d, dZ, lddz, row0);
}

bool stf_ok(int64_t c, int64_t d) { return c <= 32 && stf_lds_floats((int)c, (int)d) <= ST_MAX_LDS_FLOATS; }

#define STF_DISPATCH(CPV, ...)                          \
  switch (CPV) {                                        \
    case 4: { constexpr int CP = 4; __VA_ARGS__; } break;    \
    case 8: { constexpr int CP = 8; __VA_ARGS__; } break;    \
    case 12: { constexpr int CP = 12; __VA_ARGS__; } break;  \
    case 16: { constexpr int CP = 16; __VA_ARGS__; } break;  \
    case 20: { constexpr int CP = 20; __VA_ARGS__; } break;  \
    case 24: { constexpr int CP = 24; __VA_ARGS__; } break;  \
    case 28: { constexpr int CP = 28; __VA_ARGS__; } break;  \
    default: { constexpr int CP = 32; __VA_ARGS__; } break;  \
  }

int st_check(const char* me, int64_t n, int64_t c, int64_t d, const void* Z, int64_t ldz, const void* MU, float a) {
  if (n < 0 || c <= 0 || d <= 0) return dh::fail(DH_ERR_INVALID, "%s: bad size", me);
  if (c > ST_MAX_C || c * d > ST_MAX_CD || st_lds_floats(c, d) > ST_MAX_LDS_FLOATS)
    return dh::fail(DH_ERR_INVALID, "%s: at most %d clusters, %d cluster x feature entries and c d + 128 (c + d + 2) <= %lld (dh_student_t_supported)", me,
                    ST_MAX_C, ST_MAX_CD, (long long)ST_MAX_LDS_FLOATS);
  if (n > 0 && (!Z || !MU)) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (ldz < d) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < d", me);
  if (!(a > 0.f)) return dh::fail(DH_ERR_INVALID, "%s: the kernel's degrees-of-freedom parameter must be positive", me);
  return DH_OK;
}

}  // namespace

extern "C" int dh_student_t_supported(int64_t c, int64_t d) {
  return (c > 0 && d > 0 && c <= ST_MAX_C && c * d <= ST_MAX_CD && st_lds_floats(c, d) <= ST_MAX_LDS_FLOATS) ? 1 : 0;
}

extern "C" int dh_student_t_forward_f32(int64_t n, int64_t c, int64_t d, const float* Z, int64_t ldz, const float* MU, float a, float eps,
                                        float pw, float scale, float* Q, int64_t ldq, dh_stream_t stream) {
  const char* me = "dh_student_t_forward_f32";
  int rc = st_check(me, n, c, d, Z, ldz, MU, a);
  if (rc != DH_OK) return rc;
  if (n == 0) return DH_OK;
  if (!Q || ldq < c) return dh::fail(DH_ERR_INVALID, "%s: bad output", me);
  if (stf_ok(c, d)) {
    const size_t lds = (size_t)stf_lds_floats((int)c, (int)d) * sizeof(float);
    STF_DISPATCH(stf_cp((int)c), hipLaunchKernelGGL(student_t_forward_fast_kernel<CP>, dim3((unsigned)dh::ceil_div(n, STF_ROWS)), dim3(STF_ROWS), lds,
                                                    dh::as_stream(stream), n, (int)c, (int)d, Z, ldz, MU, StParams{a, eps, pw, scale}, Q, ldq));
    return dh::check_launch(me);
  }
  hipLaunchKernelGGL(student_t_forward_kernel, dim3((unsigned)dh::ceil_div(n, ST_ROWS)), dim3(ST_ROWS), (size_t)(c * d + ST_ROWS * (d | 1)) * sizeof(float),
                     dh::as_stream(stream), n, (int)c, (int)d, Z, ldz, MU, StParams{a, eps, pw, scale}, Q, ldq);
  return dh::check_launch(me);
}

extern "C" size_t dh_student_t_backward_workspace_bytes(int64_t n, int64_t c, int64_t d) {
  if (n <= 0 || c <= 0 || d <= 0) return 0;
  return (size_t)dh::ceil_div(n, ST_ROWS) * (size_t)(c * d + c) * sizeof(float);
}

extern "C" int dh_student_t_backward_f32(int64_t n, int64_t c, int64_t d, const float* Z, int64_t ldz, const float* MU, float a, float eps,
                                         float pw, float scale, const float* G, int64_t ldg, float* dZ, int64_t lddz, float* dMU,
                                         void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_student_t_backward_f32";
  int rc = st_check(me, n, c, d, Z, ldz, MU, a);
  if (rc != DH_OK) return rc;
  if (!dMU) return dh::fail(DH_ERR_INVALID, "%s: null dMU", me);
  hipStream_t st = dh::as_stream(stream);
  if (n == 0) {
    if (dh::zero_async(dMU, (size_t)(c * d) * sizeof(float), st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me);
    return DH_OK;
  }
  if (!G || ldg < c || (dZ && lddz < d)) return dh::fail(DH_ERR_INVALID, "%s: bad gradient operands", me);
  const size_t need = dh_student_t_backward_workspace_bytes(n, c, d);
  if (!workspace || workspace_bytes < need) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, workspace_bytes, need);
  const int64_t n_blocks = dh::ceil_div(n, ST_ROWS);
  const size_t lds = (size_t)st_lds_floats(c, d) * sizeof(float);
  if (stf_ok(c, d)) {
    const size_t lds_fast = (size_t)stf_lds_floats((int)c, (int)d) * sizeof(float);
    STF_DISPATCH(stf_cp((int)c), hipLaunchKernelGGL(student_t_backward_fast_kernel<CP>, dim3((unsigned)n_blocks), dim3(STF_ROWS), lds_fast, st, n, (int)c, (int)d,
                                                    Z, ldz, MU, StParams{a, eps, pw, scale}, G, ldg, dZ, lddz, static_cast<float*>(workspace)));
  } else
    hipLaunchKernelGGL(student_t_backward_kernel, dim3((unsigned)n_blocks), dim3(ST_ROWS), lds, st, n, (int)c, (int)d, Z, ldz, MU,
                       StParams{a, eps, pw, scale}, G, ldg, dZ, lddz, static_cast<float*>(workspace));
  rc = dh::check_launch(me);
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(student_t_reduce_kernel, dim3((unsigned)(c * d)), dim3(256), 0, st, n_blocks, (int)c, (int)d,
                     static_cast<const float*>(workspace), MU, dMU);
  return dh::check_launch("dh_student_t_backward_f32(reduce)");
}
