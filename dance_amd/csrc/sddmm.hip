// CSR SDDMM: out[e] = scale[e]? * <U[row(e), :], V[col(e), :]> for every stored edge e — the sampled dense-dense
// product of SURVEY.md §8b.  It is the edge-value gradient of the CSR SpMM (dval[e] = <dY[row(e)], Z[col(e)]>, i.e. the
// backward of the edge-weighted aggregations graphsc.py:417-426 / gnn.py:81-82 with respect to the weights) and the
// per-edge score of a sparse inner-product decoder (graphsc.py:386-411 evaluated on the stored edges only).
//
// HBM-bound gather like the SpMM: a group of G lanes owns one row, keeps its slice of U[row] in registers, streams the
// V rows of the row's edges (16-byte loads per lane, 4 edges in flight) and reduces each edge's partial products across
// the group with DPP/shuffle adds; accumulation order inside a lane is fixed (columns ascending), the cross-lane tree is
// fixed, so results are bit-reproducible.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}

// four consecutive features as fp32: one 16-byte load (f32 storage) or one 8-byte load widened exactly (bf16 storage)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<uint16_t>(const uint16_t* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  f32x4 r;
  r[0] = __uint_as_float(w.x << 16);
  r[1] = __uint_as_float(w.x & 0xffff0000u);
  r[2] = __uint_as_float(w.y << 16);
  r[3] = __uint_as_float(w.y & 0xffff0000u);
  return r;
}

// G lanes per row; each lane covers columns [4 g + 4 G a, +4) for a = 0 .. NACC-1 (width <= 4 G NACC, width % 4 == 0)
template <int G, int NACC, typename T>
__global__ __launch_bounds__(256) void sddmm_csr_kernel(int64_t n_rows, int64_t width, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, const float* __restrict__ scale,
                                                        const T* __restrict__ U, int64_t ldu, const T* __restrict__ V,
                                                        int64_t ldv, float* __restrict__ out) {
  const int g = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  if (row >= n_rows) return;
  f32x4 u[NACC];
  bool live[NACC];
  int cl[NACC];  // this lane's column of slice a, or column 0 where the slice lies beyond the width (a valid address whose value is dropped)
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int64_t c = 4 * g + 4 * G * a;
    live[a] = c < width;
    cl[a] = live[a] ? (int)c : 0;
    const f32x4 x = load4<T>(U + row * ldu + cl[a]);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[a][i] = live[a] ? x[i] : 0.f;
  }
  const int s = rowptr[row], t = rowptr[row + 1];
  // No load below sits behind a divergent guard: guarded, every load is its own exec-masked block that the compiler closes with
  // s_waitcnt vmcnt(0), and the "4 edges in flight" become 4 (x NACC) dependent round trips (ISA, round 5; the same finding as
  // gcn_narrow.hip).  Edges past the row's end repeat its last edge and are not stored.
  for (int e0 = s; e0 < t; e0 += 4) {
    int ck[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ck[k] = col[min(e0 + k, t - 1)];
    float sc[4] = {1.f, 1.f, 1.f, 1.f};
    if (scale) {  // uniform
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[k] = scale[min(e0 + k, t - 1)];
    }
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      f32x4 x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = load4<T>(V + (int64_t)ck[k] * ldv + cl[a]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[k] = fmaf(u[a][i], live[a] ? x[k][i] : 0.f, part[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = group_sum<G>(part[k]);
      if (g == 0 && e0 + k < t) out[e0 + k] = sc[k] * d;
    }
  }
}

template <typename T>
int sddmm_launch(const char* me, int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col, const float* scale,
                 const T* U, int64_t ldu, const T* V, int64_t ldv, float* out, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_rows == 0) return DH_OK;
  if (!rowptr || !col || !U || !V || !out) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  const uintptr_t align = 4 * sizeof(T);
  if (width % 4 != 0 || ldu % 4 != 0 || ldv % 4 != 0 || (uintptr_t)U % align || (uintptr_t)V % align)
    return dh::fail(DH_ERR_INVALID, "%s: rows must be multiples of 4 features and %d-byte aligned (pad the width)", me, (int)align);
  if (ldu < width || ldv < width) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < width", me);
  if (width > 4 * 64 * 8) return dh::fail(DH_ERR_INVALID, "%s: width %lld > 2048", me, (long long)width);
  hipStream_t st = dh::as_stream(stream);
  const int64_t vecs = width / 4;
#define DH_SDDMM(G, NACC)                                                                                                 \
  hipLaunchKernelGGL((sddmm_csr_kernel<G, NACC, T>), dim3((unsigned)dh::ceil_div(n_rows, 256 / G)), dim3(256), 0, st, n_rows, \
                     width, rowptr, col, scale, U, ldu, V, ldv, out)
  if (vecs > 256) DH_SDDMM(64, 8);
  else if (vecs > 128) DH_SDDMM(64, 4);
  else if (vecs > 64) DH_SDDMM(64, 2);
  else if (vecs > 32) DH_SDDMM(64, 1);
  else if (vecs > 16) DH_SDDMM(32, 1);
  else if (vecs > 8) DH_SDDMM(16, 1);
  else DH_SDDMM(8, 1);
#undef DH_SDDMM
  return dh::check_launch(me);
}

}  // namespace

extern "C" int dh_sddmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col,
                                const float* scale, const float* U, int64_t ldu, const float* V, int64_t ldv, float* out,
                                dh_stream_t stream) {
  return sddmm_launch<float>("dh_sddmm_csr_f32", n_rows, n_cols, width, rowptr, col, scale, U, ldu, V, ldv, out, stream);
}

// bf16-stored operands (config C3), fp32 products and sums, fp32 result
extern "C" int dh_sddmm_csr_bf16(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col,
                                 const float* scale, const uint16_t* U, int64_t ldu, const uint16_t* V, int64_t ldv, float* out,
                                 dh_stream_t stream) {
  return sddmm_launch<uint16_t>("dh_sddmm_csr_bf16", n_rows, n_cols, width, rowptr, col, scale, U, ldu, V, ldv, out, stream);
}
