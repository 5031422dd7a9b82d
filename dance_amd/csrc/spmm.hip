// CSR SpMM with fused epilogue — the aggregation kernel of the GCN / GraphConv layers
// (SURVEY.md §2b K1/K2/K5).  HBM-bound gather: every neighbour row of Z is a contiguous
// `width`-float read, so the kernel is organised to keep many 16-byte-per-lane row reads in
// flight per wave and to touch each Y element exactly once.
//
// Mapping: a "group" of G lanes owns one destination row (G = 64 for wide rows: one
// wavefront per row; G = 8..32 packs several short rows into one wavefront).  The group
// first loads up to G (col, val) pairs of its row with one coalesced load, then broadcasts
// them lane-by-lane (v_readlane for G = 64 so the neighbour's base address is scalar,
// ds_bpermute inside sub-wave groups) while every lane streams its VEC-wide column slice of
// the neighbour rows.  Accumulation is sequential in CSR order per output element — no
// atomics, bit-reproducible.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = f32x2; };
template <> struct VecT<4> { using type = f32x4; };

template <int VEC, typename V>
__device__ __forceinline__ void fma_vec(V& a, float w, V z) {
  if constexpr (VEC == 1) a = fmaf(w, z, a);
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] = fmaf(w, z[i], a[i]);
  }
}

__device__ __forceinline__ float epi(float a, float scale, float b, int act) {
  float y = fmaf(a, scale, b);
  return (act == DH_ACT_RELU) ? fmaxf(y, 0.f) : y;
}
template <int VEC, typename V>
__device__ __forceinline__ void epilogue(V& a, float scale, const float* bias, int64_t c, int act) {
  if constexpr (VEC == 1) a = epi(a, scale, bias ? bias[c] : 0.f, act);
  else {
    V b = bias ? *reinterpret_cast<const V*>(bias + c) : V(0.f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] = epi(a[i], scale, b[i], act);
  }
}

template <int G>
__device__ __forceinline__ int bcast_i(int v, int k) {
  if constexpr (G == 64) return __builtin_amdgcn_readlane(v, k);
  else return __shfl(v, k, G);
}
template <int G>
__device__ __forceinline__ float bcast_f(float v, int k) {
  return __int_as_float(bcast_i<G>(__float_as_int(v), k));
}

// AdaptiveSAGE per-edge factor (dance/models/nn/gnn.py:72-82): alpha[idx(e)] with idx chosen from the
// "cell_id" of the edge's source and destination node; folded into the edge weight on the fly.
struct SageScale {
  const int32_t* src_id;  // [n_src] cell_id of source nodes (>= 0 for genes, -1 for cells)
  const int32_t* dst_id;  // [n_dst]
  const float* alpha;     // [n_genes + 2]
  int n_genes;
};

__device__ __forceinline__ float sage_alpha(const SageScale& sg, int sid, int did) {
  int idx = sg.n_genes + 1;                      // cell self loop (default)
  if (sid >= 0 && did < 0) idx = sid;            // gene -> cell
  if (did >= 0 && sid < 0) idx = did;            // cell -> gene
  if (did >= 0 && sid >= 0) idx = sg.n_genes;    // gene self loop
  return sg.alpha[idx];
}

// ReLU sign masks (fusing autograd's ReluBackward into the backward gather, dh_spmm_csr_relu_f32):
// for the wavefront-per-row float4 configuration the forward pass stores, per row and per 256-column slice, the
// four wave ballots "element i of lane l's float4 is > 0" (4 x uint64 = 256 bits); the backward pass gathers
// dY rows and zeroes the elements whose bit is clear, so G = dY * [Y > 0] is never written to or read from HBM.
struct ReluMask {
  unsigned long long* out;       // forward: written when non-null   [n_rows][slices][4]
  const unsigned long long* in;  // backward: applied to the gathered rows of Z when non-null [n_cols][slices][4]
  int slices;                    // width / 256 of the whole layer
  int slice0;                    // first 256-column slice of this launch (column-sliced passes)
};

// G lanes per row, VEC floats per lane per slice, NACC slices per lane (slices G*VEC apart).
template <int G, int VEC, int NACC, bool SAGE, bool MASKED = false>
__global__ __launch_bounds__(256) void spmm_csr_kernel(
    int64_t n_rows, int64_t width, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ rowscale, const float* __restrict__ colscale,
    const float* __restrict__ Z, int64_t ldz, float* __restrict__ Y, int64_t ldy,
    const float* __restrict__ bias, int act, int reduce, SageScale sage, ReluMask mask = ReluMask{nullptr, nullptr, 0, 0}) {
  using V = typename VecT<VEC>::type;
  constexpr int ROWS_PER_BLOCK = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / G;
  if (row >= n_rows) return;  // whole groups exit together

  // this lane's first column in the current column block (blockIdx.y covers width in
  // G*VEC*NACC-wide blocks)
  const int64_t c0 = (int64_t)blockIdx.y * (G * VEC * NACC) + (int64_t)g * VEC;
  bool live[NACC];
  V acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    live[a] = (c0 + (int64_t)a * G * VEC) < width;
    acc[a] = V(0.f);
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
    // 4 neighbour rows in flight per lane per slice
    for (; k + 4 <= cnt; k += 4) {
      V z[4][NACC];
      float wk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ck = bcast_i<G>(c, k + u);
        wk[u] = bcast_f<G>(w, k + u);
        const float* zr = Z + (int64_t)ck * ldz + c0;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
          z[u][a] = live[a] ? *reinterpret_cast<const V*>(zr + a * G * VEC) : V(0.f);
        if constexpr (MASKED) {
          if (mask.in) {  // wave-uniform row ck: the 4 ballot words of each slice come through the scalar cache
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
              const unsigned long long* m = mask.in + ((int64_t)ck * mask.slices + mask.slice0 + blockIdx.y * NACC + a) * 4;
#pragma unroll
              for (int i = 0; i < VEC; ++i)
                if (!((m[i] >> g) & 1ull)) z[u][a][i] = 0.f;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < NACC; ++a) fma_vec<VEC>(acc[a], wk[u], z[u][a]);
    }
    for (; k < cnt; ++k) {
      const int ck = bcast_i<G>(c, k);
      const float wk = bcast_f<G>(w, k);
      const float* zr = Z + (int64_t)ck * ldz + c0;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        if (!live[a]) continue;
        V zv = *reinterpret_cast<const V*>(zr + a * G * VEC);
        if constexpr (MASKED) {
          if (mask.in) {
            const unsigned long long* m = mask.in + ((int64_t)ck * mask.slices + mask.slice0 + blockIdx.y * NACC + a) * 4;
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              if (!((m[i] >> g) & 1ull)) zv[i] = 0.f;
          }
        }
        fma_vec<VEC>(acc[a], wk, zv);
      }
    }
  }

  float scale = rowscale ? rowscale[row] : 1.f;
  if (reduce == DH_REDUCE_MEAN) scale = (t > s) ? scale / (float)(t - s) : 0.f;
  float* yr = Y + row * ldy + c0;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    if (!live[a]) continue;
    epilogue<VEC>(acc[a], scale, bias, c0 + (int64_t)a * G * VEC, act);
    if constexpr (MASKED) {
      if (mask.out) {
        unsigned long long* m = mask.out + (row * mask.slices + mask.slice0 + blockIdx.y * NACC + a) * 4;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const unsigned long long b = __ballot(acc[a][i] > 0.f);
          if (g == 0) m[i] = b;
        }
      }
    }
    __builtin_nontemporal_store(acc[a], reinterpret_cast<V*>(yr + a * G * VEC));
  }
}

template <int VEC, bool SAGE>
int launch_vec(int64_t n_rows, int64_t width, const int32_t* rowptr, const int32_t* col,
               const float* val, const float* rowscale, const float* colscale, const float* Z,
               int64_t ldz, float* Y, int64_t ldy, const float* bias, int act, int reduce,
               SageScale sage, hipStream_t st) {
  const int64_t vecs = dh::ceil_div(width, VEC);
#define DH_SPMM_LAUNCH(G, NACC)                                                                  \
  do {                                                                                           \
    dim3 grid((unsigned)dh::ceil_div(n_rows, 256 / G),                                           \
              (unsigned)dh::ceil_div(vecs, (int64_t)G * NACC));                                  \
    hipLaunchKernelGGL((spmm_csr_kernel<G, VEC, NACC, SAGE>), grid, dim3(256), 0, st, n_rows,    \
                       width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act,   \
                       reduce, sage);                                                            \
  } while (0)
  if (VEC == 4 && !SAGE && vecs >= 64 && vecs % 32 == 0) {
    // Wide rows: one pass per 128-column slice instead of one launch over the whole width.  Every pass gathers from a
    // working set of n_cols x 512 bytes instead of n_cols x width x 4, which the L2 / MALL hold a larger share of.
    // Measured at 1M rows, k = 15 (scripts/spmm_width_probe.py): one 512-wide launch 5.44 ms, two 256-wide passes
    // 4.90 ms, four 128-wide passes (32 lanes per row, two rows per wavefront) 4.54 ms.
    for (int64_t c = 0; c < width; c += 128) {
      dim3 grid((unsigned)dh::ceil_div(n_rows, 8), 1);
      hipLaunchKernelGGL((spmm_csr_kernel<32, VEC, 1, SAGE>), grid, dim3(256), 0, st, n_rows, (int64_t)128, rowptr, col, val,
                         rowscale, colscale, Z + c, ldz, Y + c, ldy, bias ? bias + c : nullptr, act, reduce, sage);
    }
  } else if (vecs > 64) DH_SPMM_LAUNCH(64, 2);
  else if (vecs > 32) DH_SPMM_LAUNCH(64, 1);
  else if (vecs > 16) DH_SPMM_LAUNCH(32, 1);
  else if (vecs > 8) DH_SPMM_LAUNCH(16, 1);
  else DH_SPMM_LAUNCH(8, 1);
#undef DH_SPMM_LAUNCH
  return dh::check_launch(SAGE ? "dh_sage_aggregate_f32" : "dh_spmm_csr_f32");
}

template <bool SAGE>
int dispatch(int64_t n_rows, int64_t width, const int32_t* rowptr, const int32_t* col, const float* val,
             const float* rowscale, const float* colscale, const float* Z, int64_t ldz, float* Y,
             int64_t ldy, const float* bias, int act, int reduce, SageScale sage, hipStream_t st) {
  const bool a16 = dh::aligned16(Z) && dh::aligned16(Y) && (!bias || dh::aligned16(bias));
  const bool a8 = ((uintptr_t)Z % 8 == 0) && ((uintptr_t)Y % 8 == 0) && (!bias || (uintptr_t)bias % 8 == 0);
  if (a16 && width % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0)
    return launch_vec<4, SAGE>(n_rows, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage, st);
  if (a8 && width % 2 == 0 && ldz % 2 == 0 && ldy % 2 == 0)
    return launch_vec<2, SAGE>(n_rows, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage, st);
  return launch_vec<1, SAGE>(n_rows, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage, st);
}

}  // namespace

extern "C" int dh_spmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t width,
                               const int32_t* rowptr, const int32_t* col, const float* val,
                               const float* rowscale, const float* colscale, const float* Z,
                               int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                               int reduce, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!rowptr || !Z || !Y) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: null rowptr/Z/Y");
  if (ldz < width || ldy < width) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: leading dimension < width");
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: bad act %d", act);
  if (reduce != DH_REDUCE_SUM && reduce != DH_REDUCE_MEAN) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: bad reduce %d", reduce);
  if (n_rows >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: n_rows >= 2^31");
  return dispatch<false>(n_rows, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce,
                         SageScale{nullptr, nullptr, nullptr, 0}, dh::as_stream(stream));
}

extern "C" size_t dh_relu_mask_bytes(int64_t n_rows, int64_t width) {
  if (n_rows <= 0 || width <= 0 || width % 256 != 0) return 0;
  return (size_t)n_rows * (size_t)(width / 256) * 4 * sizeof(unsigned long long);
}

// SpMM with the ReLU of a GCN layer fused on both sides (see ReluMask): forward = dh_spmm_csr_f32(act = relu)
// that additionally records the sign mask of its output; backward = SpMM whose gathered operand is dY with the
// recorded mask applied on the fly.  Only for the float4 wavefront-per-row configuration: width % 256 == 0 and
// 16-byte aligned operands (the host falls back to dh_relu_backward_f32 + dh_spmm_csr_f32 otherwise).
extern "C" int dh_spmm_csr_relu_f32(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr,
                                    const int32_t* col, const float* val, const float* Z, int64_t ldz, float* Y,
                                    int64_t ldy, const float* bias, int act, void* out_mask, const void* in_mask,
                                    dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!rowptr || !Z || !Y) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: null rowptr/Z/Y");
  if (width % 256 != 0 || ldz % 4 != 0 || ldy % 4 != 0 || !dh::aligned16(Z) || !dh::aligned16(Y) || (bias && !dh::aligned16(bias)))
    return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: needs width %% 256 == 0 and 16-byte aligned rows");
  if (ldz < width || ldy < width) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: leading dimension < width");
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: bad act %d", act);
  if (n_rows >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: n_rows >= 2^31");
  const SageScale none{nullptr, nullptr, nullptr, 0};
  hipStream_t st = dh::as_stream(stream);
  // one pass per 256-column slice (see launch_vec): the gathered working set of a pass is n_cols x 1 KB
  for (int64_t c = 0; c < width; c += 256) {
    const ReluMask mask{static_cast<unsigned long long*>(out_mask), static_cast<const unsigned long long*>(in_mask), (int)(width / 256),
                        (int)(c / 256)};
    dim3 grid((unsigned)dh::ceil_div(n_rows, 4), 1);
    hipLaunchKernelGGL((spmm_csr_kernel<64, 4, 1, false, true>), grid, dim3(256), 0, st, n_rows, (int64_t)256, rowptr, col, val, nullptr,
                       nullptr, Z + c, ldz, Y + c, ldy, bias ? bias + c : nullptr, act, DH_REDUCE_SUM, none, mask);
  }
  return dh::check_launch("dh_spmm_csr_relu_f32");
}

extern "C" int dh_sage_aggregate_f32(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes,
                                     const int32_t* rowptr, const int32_t* col, const float* w,
                                     const int32_t* src_cell_id, const int32_t* dst_cell_id,
                                     const float* alpha, const float* H, int64_t ldh, float* neigh,
                                     int64_t ldn, dh_stream_t stream) {
  if (n_dst < 0 || n_src < 0 || width < 0 || n_genes < 0) return dh::fail(DH_ERR_INVALID, "dh_sage_aggregate_f32: negative size");
  if (n_dst == 0 || width == 0) return DH_OK;
  if (!rowptr || !H || !neigh || !src_cell_id || !dst_cell_id || !alpha)
    return dh::fail(DH_ERR_INVALID, "dh_sage_aggregate_f32: null pointer");
  if (ldh < width || ldn < width) return dh::fail(DH_ERR_INVALID, "dh_sage_aggregate_f32: leading dimension < width");
  return dispatch<true>(n_dst, width, rowptr, col, w, nullptr, nullptr, H, ldh, neigh, ldn, nullptr, DH_ACT_NONE,
                        DH_REDUCE_MEAN, SageScale{src_cell_id, dst_cell_id, alpha, (int)n_genes}, dh::as_stream(stream));
}
