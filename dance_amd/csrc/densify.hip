// Densified-operand form of the AdaptiveSAGE aggregation (dance/models/nn/gnn.py:62-90).
//
// The cell-gene graph is 10 % dense (200 expressed genes of 2000 per cell).  On the vector ALUs the aggregation is bound by
// operand delivery, not by HBM (an LDS-staged gather measured no faster than the L2 gather in round 2: 13.8 vs 13.5 ms, profiles/r02_sage_lds_pmc.json).  The matrix cores are
// 16x faster than the vector ALUs, so at 10 % density a DENSE product wins: write the weighted adjacency window as a dense
// bf16 (or fp32) matrix once (one streaming pass, HBM-bound) and let the MFMA GEMM do  neigh = A_dense * H  — for the
// cell <- gene direction (A: cells x genes) and, even more so, for the gene <- cell direction (A: genes x cells), whose
// 1e5-edge rows serialise the per-row gather kernel.  Entries the window does not cover (the self loops) are added by a
// small gather kernel (dh_sage_tail) that initialises the GEMM's accumulate-into output.
//
//   dh_csr_densify_window : out[r][c - c0] = val[e] * rowscale[r] * colscale[c - c0] (* 1/deg(r) if mean) for the edges of
//                           row r whose column lies in [c0, c0 + n_cols); every other entry of the row is written as 0.
//                           Narrow windows (<= 16384 columns) build each row in LDS (one wavefront per row: zero, scatter,
//                           coalesced copy-out — the matrix is written exactly once); wide windows are zero-filled and
//                           scattered into directly.  A column that occurs twice in a row is summed in the narrow path and
//                           last-write-wins in the wide path (the cell-gene graph has no duplicates).
//   dh_sage_tail          : neigh[v,:] = 1/deg(v) * sum over the edges of v whose source lies OUTSIDE the window of
//                           alpha[idx(e)] w_e H[u,:]   (idx as in dh_sage_aggregate_f32; deg = all in-edges of v).
#include "common.h"

namespace {

__device__ __forceinline__ unsigned int f32_to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// one wavefront per row; the row is assembled in this wavefront's LDS strip [n_cols] fp32
template <bool BF16>
__global__ __launch_bounds__(256) void densify_rows_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const float* __restrict__ val, const float* __restrict__ rowscale,
                                                           const float* __restrict__ colscale, int mean, int col_begin, int n_cols,
                                                           void* __restrict__ outv, int64_t ldo, int waves) {
  extern __shared__ __attribute__((aligned(16))) float strip[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= waves) return;
  float* buf = strip + (size_t)wave * n_cols;
  for (int64_t row = (int64_t)blockIdx.x * waves + wave; row < n_rows; row += (int64_t)gridDim.x * waves) {
    for (int c = lane; c < n_cols; c += 64) buf[c] = 0.f;
    const int s = rowptr[row], t = rowptr[row + 1];
    float f = rowscale ? rowscale[row] : 1.f;
    if (mean) f = (t > s) ? f / (float)(t - s) : 0.f;
    for (int e = s + lane; e < t; e += 64) {
      const unsigned int c = (unsigned int)(col[e] - col_begin);
      if (c < (unsigned int)n_cols) atomicAdd(&buf[c], (val ? val[e] : 1.f) * f * (colscale ? colscale[c] : 1.f));  // ds_add_f32
    }
    // the wavefront's own LDS traffic is ordered: no barrier needed
    if (BF16) {
      uint16_t* o = static_cast<uint16_t*>(outv) + row * ldo;
      for (int c = 2 * lane; c < n_cols; c += 128) {
        if (c + 1 < n_cols && (((uintptr_t)(o + c)) & 3u) == 0) *reinterpret_cast<unsigned int*>(o + c) = f32_to_bf16(buf[c]) | (f32_to_bf16(buf[c + 1]) << 16);
        else {
          o[c] = (uint16_t)f32_to_bf16(buf[c]);
          if (c + 1 < n_cols) o[c + 1] = (uint16_t)f32_to_bf16(buf[c + 1]);
        }
      }
    } else {
      float* o = static_cast<float*>(outv) + row * ldo;
      for (int c = lane; c < n_cols; c += 64) o[c] = buf[c];
    }
  }
}

// wide windows: the output has been zero-filled; one wavefront per (row, chunk of 4096 edges)
template <bool BF16>
__global__ __launch_bounds__(256) void densify_scatter_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              const float* __restrict__ val, const float* __restrict__ rowscale,
                                                              const float* __restrict__ colscale, int mean, int64_t col_begin, int64_t n_cols,
                                                              void* __restrict__ outv, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.y;
  const int s = rowptr[row], t = rowptr[row + 1];
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t e0 = (int64_t)s + chunk * 4096;
  if (e0 >= t) return;
  const int64_t e1 = min((int64_t)t, e0 + 4096);
  float f = rowscale ? rowscale[row] : 1.f;
  if (mean) f = f / (float)(t - s);
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    const int64_t c = (int64_t)col[e] - col_begin;
    if (c < 0 || c >= n_cols) continue;
    const float v = (val ? val[e] : 1.f) * f * (colscale ? colscale[c] : 1.f);
    if (BF16) static_cast<uint16_t*>(outv)[row * ldo + c] = (uint16_t)f32_to_bf16(v);
    else static_cast<float*>(outv)[row * ldo + c] = v;
  }
}

__device__ __forceinline__ float sage_alpha_idx(const float* __restrict__ alpha, int n_genes, int sid, int did) {
  int idx = n_genes + 1;
  if (sid >= 0 && did < 0) idx = sid;
  if (did >= 0 && sid < 0) idx = did;
  if (did >= 0 && sid >= 0) idx = n_genes;
  return alpha[idx];
}

// one wavefront per destination row; lanes own column pairs (strided over the width); edges outside the window only
template <bool HBF16, bool OBF16>
__global__ __launch_bounds__(256) void sage_tail_kernel(int64_t n_dst, int64_t width, int n_genes, int64_t col_begin, int64_t n_cols,
                                                        const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const float* __restrict__ w, const int32_t* __restrict__ src_id,
                                                        const int32_t* __restrict__ dst_id, const float* __restrict__ alpha,
                                                        const void* __restrict__ Hv, int64_t ldh, void* __restrict__ outv, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_dst) return;
  const int s = rowptr[row], t = rowptr[row + 1];
  const int did = dst_id[row];
  const float inv = (t > s) ? 1.f / (float)(t - s) : 0.f;
  for (int64_t c0 = 0; c0 < width; c0 += 128) {
    const int64_t c = c0 + 2 * lane;
    float ax = 0.f, ay = 0.f;
    for (int base = s; base < t; base += 64) {
      const int e = base + lane;
      int ce = -1;
      float fe = 0.f;
      if (e < t) {
        const int cc = col[e];
        if ((int64_t)cc < col_begin || (int64_t)cc >= col_begin + n_cols) {
          ce = cc;
          fe = w[e] * sage_alpha_idx(alpha, n_genes, src_id[cc], did);
        }
      }
      unsigned long long m = __ballot(ce >= 0);
      while (m) {  // the few out-of-window edges of this 64-edge stretch, in CSR order
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int ck = __builtin_amdgcn_readlane(ce, k);
        const float fk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fe), k));
        if (c < width) {
          if (HBF16) {
            const uint16_t* h = static_cast<const uint16_t*>(Hv) + (int64_t)ck * ldh + c;
            ax = fmaf(fk, __uint_as_float((unsigned int)h[0] << 16), ax);
            if (c + 1 < width) ay = fmaf(fk, __uint_as_float((unsigned int)h[1] << 16), ay);
          } else {
            const float* h = static_cast<const float*>(Hv) + (int64_t)ck * ldh + c;
            ax = fmaf(fk, h[0], ax);
            if (c + 1 < width) ay = fmaf(fk, h[1], ay);
          }
        }
      }
    }
    if (c < width) {
      if (OBF16) {
        uint16_t* o = static_cast<uint16_t*>(outv) + row * ldo + c;
        o[0] = (uint16_t)f32_to_bf16(ax * inv);
        if (c + 1 < width) o[1] = (uint16_t)f32_to_bf16(ay * inv);
      } else {
        float* o = static_cast<float*>(outv) + row * ldo + c;
        o[0] = ax * inv;
        if (c + 1 < width) o[1] = ay * inv;
      }
    }
  }
}

}  // namespace

extern "C" int dh_csr_densify_window(int64_t n_rows, int64_t max_row_nnz, const int32_t* rowptr, const int32_t* col, const float* val,
                                     const float* rowscale, const float* colscale, int mean, int64_t col_begin, int64_t n_cols,
                                     void* out, int64_t ldo, int out_dtype, dh_stream_t stream) {
  const char* me = "dh_csr_densify_window";
  if (n_rows < 0 || n_cols < 0 || col_begin < 0 || max_row_nnz < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!rowptr || !col || !out) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (ldo < n_cols) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < window width", me);
  if (out_dtype != DH_DTYPE_F32 && out_dtype != DH_DTYPE_BF16) return dh::fail(DH_ERR_INVALID, "%s: bad dtype", me);
  if (col_begin + n_cols >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: window beyond int32 columns", me);
  hipStream_t st = dh::as_stream(stream);
  const bool bf16 = out_dtype == DH_DTYPE_BF16;
  if (n_cols <= 16384) {
    const int waves = (int)(n_cols <= 4096 ? 4 : n_cols <= 8192 ? 2 : 1);
    const size_t lds = (size_t)waves * n_cols * sizeof(float);
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(densify_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess &&
                           hipFuncSetAttribute(reinterpret_cast<const void*>(densify_rows_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess;
    if (!ok) return dh::fail(DH_ERR_LAUNCH, "%s: cannot reserve LDS", me);
    const int64_t blocks = dh::ceil_div(n_rows, waves);
    const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);
    if (bf16) hipLaunchKernelGGL(densify_rows_kernel<true>, dim3(grid), dim3(256), lds, st, n_rows, rowptr, col, val, rowscale, colscale, mean, (int)col_begin, (int)n_cols, out, ldo, waves);
    else hipLaunchKernelGGL(densify_rows_kernel<false>, dim3(grid), dim3(256), lds, st, n_rows, rowptr, col, val, rowscale, colscale, mean, (int)col_begin, (int)n_cols, out, ldo, waves);
    return dh::check_launch(me);
  }
  if (n_rows > 65535) return dh::fail(DH_ERR_INVALID, "%s: windows wider than 16384 columns support at most 65535 rows", me);
  const size_t esz = bf16 ? 2 : 4;
  if (ldo == n_cols) { if (dh::zero_async(out, (size_t)n_rows * n_cols * esz, st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me); }
  else if (dh::zero2d_async(out, (size_t)ldo * esz, (size_t)n_cols * esz, (size_t)n_rows, st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me);
  const int64_t chunks = dh::ceil_div(max_row_nnz > 0 ? max_row_nnz : 1, 4096);
  dim3 grid((unsigned)dh::ceil_div(chunks, 4), (unsigned)n_rows);
  if (bf16) hipLaunchKernelGGL(densify_scatter_kernel<true>, grid, dim3(256), 0, st, n_rows, rowptr, col, val, rowscale, colscale, mean, col_begin, n_cols, out, ldo);
  else hipLaunchKernelGGL(densify_scatter_kernel<false>, grid, dim3(256), 0, st, n_rows, rowptr, col, val, rowscale, colscale, mean, col_begin, n_cols, out, ldo);
  return dh::check_launch(me);
}

extern "C" int dh_sage_tail(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes, int64_t col_begin, int64_t n_cols,
                            const int32_t* rowptr, const int32_t* col, const float* w, const int32_t* src_cell_id,
                            const int32_t* dst_cell_id, const float* alpha, const void* H, int64_t ldh, int h_dtype, void* neigh,
                            int64_t ldn, int out_dtype, dh_stream_t stream) {
  const char* me = "dh_sage_tail";
  if (n_dst < 0 || n_src < 0 || width < 0 || n_genes < 0 || col_begin < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_dst == 0 || width == 0) return DH_OK;
  if (!rowptr || !col || !w || !src_cell_id || !dst_cell_id || !alpha || !H || !neigh) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (ldh < width || ldn < width) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < width", me);
  if ((h_dtype != DH_DTYPE_F32 && h_dtype != DH_DTYPE_BF16) || (out_dtype != DH_DTYPE_F32 && out_dtype != DH_DTYPE_BF16)) return dh::fail(DH_ERR_INVALID, "%s: bad dtype", me);
  hipStream_t st = dh::as_stream(stream);
  dim3 grid((unsigned)dh::ceil_div(n_dst, 4));
#define DH_TAIL(HB, OB) hipLaunchKernelGGL((sage_tail_kernel<HB, OB>), grid, dim3(256), 0, st, n_dst, width, (int)n_genes, col_begin, n_cols, rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, ldh, neigh, ldn)
  if (h_dtype == DH_DTYPE_BF16 && out_dtype == DH_DTYPE_BF16) DH_TAIL(true, true);
  else if (h_dtype == DH_DTYPE_BF16) DH_TAIL(true, false);
  else if (out_dtype == DH_DTYPE_BF16) DH_TAIL(false, true);
  else DH_TAIL(false, false);
#undef DH_TAIL
  return dh::check_launch(me);
}
