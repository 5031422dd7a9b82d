// CSR SpMM over bf16-stored features (bf16 storage, fp32 accumulation).
//
// Same organisation as spmm.hip — a group of G lanes owns one destination row, the row's (col, val) pairs are
// loaded once and broadcast, accumulation is sequential in CSR order with one fmaf per term (bit-reproducible,
// no atomics) — but a lane's 16-byte slice now carries 8 bf16 features, so a gathered neighbour row costs half the
// HBM bytes of the fp32 kernel.  Each bf16 is widened exactly (bits << 16) and all arithmetic is fp32; the output
// is written as fp32, or rounded once (round-to-nearest-even) to bf16.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }

// round-to-nearest-even f32 -> bf16 (NaN stays a quiet NaN)
__device__ __forceinline__ unsigned int f32_to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) { return f32_to_bf16(lo) | (f32_to_bf16(hi) << 16); }

template <int G>
__device__ __forceinline__ int bcast_i(int v, int k) {
  if constexpr (G == 64) return __builtin_amdgcn_readlane(v, k);
  else return __shfl(v, k, G);
}
template <int G>
__device__ __forceinline__ float bcast_f(float v, int k) {
  return __int_as_float(bcast_i<G>(__float_as_int(v), k));
}

// AdaptiveSAGE per-edge factor, as in spmm.hip (dance/models/nn/gnn.py:72-82)
struct SageScale {
  const int32_t* src_id;
  const int32_t* dst_id;
  const float* alpha;
  int n_genes;
};
__device__ __forceinline__ float sage_alpha(const SageScale& sg, int sid, int did) {
  int idx = sg.n_genes + 1;
  if (sid >= 0 && did < 0) idx = sid;
  if (did >= 0 && sid < 0) idx = did;
  if (did >= 0 && sid >= 0) idx = sg.n_genes;
  return sg.alpha[idx];
}

__device__ __forceinline__ void fma8(float (&a)[8], float w, u32x4 z) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[2 * i] = fmaf(w, bf16_lo(z[i]), a[2 * i]);
    a[2 * i + 1] = fmaf(w, bf16_hi(z[i]), a[2 * i + 1]);
  }
}

// G lanes per row, 8 bf16 per lane per slice, NACC slices per lane (slices G*8 columns apart).
template <int G, int NACC, bool SAGE, bool OUT_BF16>
__global__ __launch_bounds__(256) void spmm_csr_bf16_kernel(
    int64_t n_rows, int64_t width, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ rowscale, const float* __restrict__ colscale,
    const uint16_t* __restrict__ Z, int64_t ldz, void* __restrict__ Yv, int64_t ldy, const float* __restrict__ bias,
    int act, int reduce, SageScale sage) {
  constexpr int ROWS_PER_BLOCK = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / G;
  if (row >= n_rows) return;  // whole groups exit together
  const int64_t c0 = (int64_t)blockIdx.y * (G * 8 * NACC) + (int64_t)g * 8;
  bool live[NACC];
  float acc[NACC][8];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    live[a] = (c0 + (int64_t)a * G * 8) < width;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[a][i] = 0.f;
  }

  const int s = rowptr[row], t = rowptr[row + 1];
  for (int base = s; base < t; base += G) {
    const int e = base + g;
    int c = 0;
    float w = 0.f;
    if (e < t) {
      c = col[e];
      w = val ? val[e] : 1.f;
      if (colscale) w *= colscale[c];
      if constexpr (SAGE) w *= sage_alpha(sage, sage.src_id[c], sage.dst_id[row]);
    }
    const int cnt = min(G, t - base);
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {  // 4 neighbour rows in flight per lane per slice
      u32x4 z[4][NACC];
      float wk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ck = bcast_i<G>(c, k + u);
        wk[u] = bcast_f<G>(w, k + u);
        const uint16_t* zr = Z + (int64_t)ck * ldz + c0;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
          z[u][a] = live[a] ? *reinterpret_cast<const u32x4*>(zr + a * G * 8) : u32x4(0u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < NACC; ++a) fma8(acc[a], wk[u], z[u][a]);
    }
    for (; k < cnt; ++k) {
      const int ck = bcast_i<G>(c, k);
      const float wk = bcast_f<G>(w, k);
      const uint16_t* zr = Z + (int64_t)ck * ldz + c0;
#pragma unroll
      for (int a = 0; a < NACC; ++a)
        if (live[a]) fma8(acc[a], wk, *reinterpret_cast<const u32x4*>(zr + a * G * 8));
    }
  }

  float scale = rowscale ? rowscale[row] : 1.f;
  if (reduce == DH_REDUCE_MEAN) scale = (t > s) ? scale / (float)(t - s) : 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    if (!live[a]) continue;
    const int64_t c = c0 + (int64_t)a * G * 8;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      y[i] = fmaf(acc[a][i], scale, bias ? bias[c + i] : 0.f);
      if (act == DH_ACT_RELU) y[i] = fmaxf(y[i], 0.f);
    }
    if constexpr (OUT_BF16) {
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = pack_bf16(y[2 * i], y[2 * i + 1]);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(static_cast<uint16_t*>(Yv) + row * ldy + c));
    } else {
      float* yr = static_cast<float*>(Yv) + row * ldy + c;
      __builtin_nontemporal_store(f32x4{y[0], y[1], y[2], y[3]}, reinterpret_cast<f32x4*>(yr));
      __builtin_nontemporal_store(f32x4{y[4], y[5], y[6], y[7]}, reinterpret_cast<f32x4*>(yr + 4));
    }
  }
}

template <bool SAGE, bool OUT_BF16>
int launch(int64_t n_rows, int64_t width, const int32_t* rowptr, const int32_t* col, const float* val,
           const float* rowscale, const float* colscale, const uint16_t* Z, int64_t ldz, void* Y, int64_t ldy,
           const float* bias, int act, int reduce, SageScale sage, hipStream_t st, const char* what) {
  const int64_t vecs = width / 8;
#define DH_SPMM_BF16(G, NACC)                                                                                      \
  do {                                                                                                             \
    dim3 grid((unsigned)dh::ceil_div(n_rows, 256 / G), (unsigned)dh::ceil_div(vecs, (int64_t)G * NACC));           \
    hipLaunchKernelGGL((spmm_csr_bf16_kernel<G, NACC, SAGE, OUT_BF16>), grid, dim3(256), 0, st, n_rows, width,     \
                       rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage);             \
  } while (0)
  if (vecs > 64) DH_SPMM_BF16(64, 2);
  else if (vecs > 32) DH_SPMM_BF16(64, 1);
  else if (vecs > 16) DH_SPMM_BF16(32, 1);
  else if (vecs > 8) DH_SPMM_BF16(16, 1);
  else DH_SPMM_BF16(8, 1);
#undef DH_SPMM_BF16
  return dh::check_launch(what);
}

int check_common(const char* what, int64_t n_rows, int64_t n_cols, int64_t width, const void* rowptr, const void* Z,
                 int64_t ldz, const void* Y, int64_t ldy, int y_dtype, const float* bias) {
  if (n_rows < 0 || n_cols < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", what);
  if (!rowptr || !Z || !Y) return dh::fail(DH_ERR_INVALID, "%s: null rowptr/Z/Y", what);
  if (y_dtype != DH_DTYPE_F32 && y_dtype != DH_DTYPE_BF16) return dh::fail(DH_ERR_INVALID, "%s: bad output dtype %d", what, y_dtype);
  if (width % 8 != 0 || ldz % 8 != 0 || ldy % 8 != 0 || !dh::aligned16(Z) || !dh::aligned16(Y) || (bias && !dh::aligned16(bias)))
    return dh::fail(DH_ERR_INVALID, "%s: bf16 rows must be multiples of 8 features and 16-byte aligned (pad the feature width)", what);
  if (ldz < width || ldy < width) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < width", what);
  if (n_rows >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: n_rows >= 2^31", what);
  return DH_OK;
}

}  // namespace

extern "C" int dh_spmm_csr_bf16(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col,
                                const float* val, const float* rowscale, const float* colscale, const uint16_t* Z,
                                int64_t ldz, void* Y, int64_t ldy, int y_dtype, const float* bias, int act, int reduce,
                                dh_stream_t stream) {
  if (n_rows == 0 || width == 0) return DH_OK;
  if (int rc = check_common("dh_spmm_csr_bf16", n_rows, n_cols, width, rowptr, Z, ldz, Y, ldy, y_dtype, bias)) return rc;
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_bf16: bad act %d", act);
  if (reduce != DH_REDUCE_SUM && reduce != DH_REDUCE_MEAN) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_bf16: bad reduce %d", reduce);
  const SageScale none{nullptr, nullptr, nullptr, 0};
  hipStream_t st = dh::as_stream(stream);
  if (y_dtype == DH_DTYPE_BF16)
    return launch<false, true>(n_rows, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, none, st, "dh_spmm_csr_bf16");
  return launch<false, false>(n_rows, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, none, st, "dh_spmm_csr_bf16");
}

extern "C" int dh_sage_aggregate_bf16(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes, const int32_t* rowptr,
                                      const int32_t* col, const float* w, const int32_t* src_cell_id,
                                      const int32_t* dst_cell_id, const float* alpha, const uint16_t* H, int64_t ldh,
                                      void* neigh, int64_t ldn, int neigh_dtype, dh_stream_t stream) {
  if (n_dst == 0 || width == 0) return DH_OK;
  if (n_genes < 0) return dh::fail(DH_ERR_INVALID, "dh_sage_aggregate_bf16: negative size");
  if (int rc = check_common("dh_sage_aggregate_bf16", n_dst, n_src, width, rowptr, H, ldh, neigh, ldn, neigh_dtype, nullptr)) return rc;
  if (!src_cell_id || !dst_cell_id || !alpha) return dh::fail(DH_ERR_INVALID, "dh_sage_aggregate_bf16: null pointer");
  const SageScale sg{src_cell_id, dst_cell_id, alpha, (int)n_genes};
  hipStream_t st = dh::as_stream(stream);
  if (neigh_dtype == DH_DTYPE_BF16)
    return launch<true, true>(n_dst, width, rowptr, col, w, nullptr, nullptr, H, ldh, neigh, ldn, nullptr, DH_ACT_NONE, DH_REDUCE_MEAN, sg, st, "dh_sage_aggregate_bf16");
  return launch<true, false>(n_dst, width, rowptr, col, w, nullptr, nullptr, H, ldh, neigh, ldn, nullptr, DH_ACT_NONE, DH_REDUCE_MEAN, sg, st, "dh_sage_aggregate_bf16");
}
