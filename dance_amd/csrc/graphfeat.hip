// Cell-gene graph kernels: edge normalisation of CellFeatureGraph and the
// alpha gradient of AdaptiveSAGE.
#include "common.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One wavefront per row: val[e] <- deg * val[e] / sum(val[row])  (cell_feature_graph.py:62-68).
// Row sums are accumulated in f64 (rows of gene nodes can hold ~1e6 edges) and rounded to f32 once.
__global__ __launch_bounds__(256) void row_normalize_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr,
                                                            const float* __restrict__ val, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int s = rowptr[row], t = rowptr[row + 1];
  if (t == s) return;
  double acc = 0.0;
  for (int e = s + lane; e < t; e += 64) acc += (double)val[e];
  const float sum = (float)wave_sum(acc);
  const float deg = (float)(t - s);
  for (int e = s + lane; e < t; e += 64) out[e] = __fdiv_rn(__fmul_rn(deg, val[e]), sum);
}

// dalpha[idx(e)] += w_e * <H[u], dneigh[v]> / deg(v); one wavefront per destination row.
__global__ __launch_bounds__(256) void sage_alpha_grad_kernel(int64_t n_dst, int64_t width, int n_genes,
                                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              const float* __restrict__ w, const int32_t* __restrict__ src_id,
                                                              const int32_t* __restrict__ dst_id, const float* __restrict__ H,
                                                              int64_t ldh, const float* __restrict__ dneigh, int64_t ldn,
                                                              float* __restrict__ dalpha) {
  __shared__ float self_bins[2];
  if (threadIdx.x < 2) self_bins[threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < n_dst) {
    const int s = rowptr[row], t = rowptr[row + 1];
    const int did = dst_id[row];
    const float inv_deg = t > s ? 1.f / (float)(t - s) : 0.f;
    const float* g = dneigh + row * ldn;
    for (int e = s; e < t; ++e) {
      const int c = col[e];
      const float* h = H + (int64_t)c * ldh;
      float dot = 0.f;
      for (int64_t j = lane; j < width; j += 64) dot = fmaf(h[j], g[j], dot);
      dot = wave_sum(dot);
      if (lane == 0) {
        const int sid = src_id[c];
        int idx = n_genes + 1;
        if (sid >= 0 && did < 0) idx = sid;
        if (did >= 0 && sid < 0) idx = did;
        if (did >= 0 && sid >= 0) idx = n_genes;
        const float v = w[e] * dot * inv_deg;
        if (idx >= n_genes) atomicAdd(&self_bins[idx - n_genes], v);  // per-block pre-reduction of the two hot bins
        else atomicAdd(&dalpha[idx], v);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 && self_bins[threadIdx.x] != 0.f) atomicAdd(&dalpha[n_genes + threadIdx.x], self_bins[threadIdx.x]);
}

}  // namespace

extern "C" int dh_csr_row_normalize_f32(int64_t n_rows, const int32_t* rowptr, const float* val, float* out_val,
                                        dh_stream_t stream) {
  if (n_rows < 0) return dh::fail(DH_ERR_INVALID, "dh_csr_row_normalize_f32: negative size");
  if (n_rows == 0) return DH_OK;
  if (!rowptr || !val || !out_val) return dh::fail(DH_ERR_INVALID, "dh_csr_row_normalize_f32: null pointer");
  hipLaunchKernelGGL(row_normalize_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream),
                     n_rows, rowptr, val, out_val);
  return dh::check_launch("dh_csr_row_normalize_f32");
}

extern "C" int dh_sage_alpha_grad_f32(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes,
                                      const int32_t* rowptr, const int32_t* col, const float* w,
                                      const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* H,
                                      int64_t ldh, const float* dneigh, int64_t ldn, float* dalpha,
                                      dh_stream_t stream) {
  if (n_dst < 0 || n_src < 0 || width < 0 || n_genes < 0) return dh::fail(DH_ERR_INVALID, "dh_sage_alpha_grad_f32: negative size");
  if (!dalpha) return dh::fail(DH_ERR_INVALID, "dh_sage_alpha_grad_f32: null dalpha");
  hipStream_t st = dh::as_stream(stream);
  if (dh::zero_async(dalpha, (size_t)(n_genes + 2) * sizeof(float), st) != hipSuccess)
    return dh::fail(DH_ERR_LAUNCH, "dh_sage_alpha_grad_f32: memset failed");
  if (n_dst == 0 || width == 0) return DH_OK;
  if (!rowptr || !col || !w || !src_cell_id || !dst_cell_id || !H || !dneigh)
    return dh::fail(DH_ERR_INVALID, "dh_sage_alpha_grad_f32: null pointer");
  hipLaunchKernelGGL(sage_alpha_grad_kernel, dim3((unsigned)dh::ceil_div(n_dst, 4)), dim3(256), 0, st, n_dst, width,
                     (int)n_genes, rowptr, col, w, src_cell_id, dst_cell_id, H, ldh, dneigh, ldn, dalpha);
  return dh::check_launch("dh_sage_alpha_grad_f32");
}

// ---- CellFeatureGraph assembly (cell_feature_graph.py:38-69) in CSR-by-destination form ------------------------
// Nodes: genes [0,G), cells [G,G+N).  Row of gene g = its cell->gene in-edges (reference edge id = position of the
// entry in row-major nonzero order of X) followed by its self loop; row of cell c = its gene->cell in-edges
// (edge id nnz + position) followed by its self loop (edge ids 2nnz + node).  `eid` keeps the reference's edge
// order recoverable bit-exactly.
namespace {
__global__ __launch_bounds__(256) void cellgene_assemble_kernel(
    int64_t n_cells, int64_t n_genes, int64_t nnz, const int32_t* __restrict__ rowptr_x, const int32_t* __restrict__ col_x,
    const float* __restrict__ val_x, const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
    const float* __restrict__ val_t, const int32_t* __restrict__ perm_t, int32_t* __restrict__ rowptr,
    int32_t* __restrict__ col, float* __restrict__ val, int32_t* __restrict__ eid) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_nodes = n_genes + n_cells;
  if (row > n_nodes) return;
  if (row == n_nodes) {
    if (lane == 0) rowptr[row] = (int32_t)(2 * nnz + n_nodes);
    return;
  }
  if (row < n_genes) {
    const int s = rowptr_t[row], t = rowptr_t[row + 1];
    const int64_t base = (int64_t)s + row;
    if (lane == 0) rowptr[row] = (int32_t)base;
    for (int e = s + lane; e < t; e += 64) {
      col[base + (e - s)] = (int32_t)(n_genes + col_t[e]);
      val[base + (e - s)] = val_t[e];
      eid[base + (e - s)] = perm_t[e];
    }
    if (lane == 0) {
      col[base + (t - s)] = (int32_t)row;
      val[base + (t - s)] = 1.f;
      eid[base + (t - s)] = (int32_t)(2 * nnz + row);
    }
  } else {
    const int64_t c = row - n_genes;
    const int s = rowptr_x[c], t = rowptr_x[c + 1];
    const int64_t base = nnz + n_genes + (int64_t)s + c;
    if (lane == 0) rowptr[row] = (int32_t)base;
    for (int e = s + lane; e < t; e += 64) {
      col[base + (e - s)] = col_x[e];
      val[base + (e - s)] = val_x[e];
      eid[base + (e - s)] = (int32_t)(nnz + e);
    }
    if (lane == 0) {
      col[base + (t - s)] = (int32_t)row;
      val[base + (t - s)] = 1.f;
      eid[base + (t - s)] = (int32_t)(2 * nnz + row);
    }
  }
}
}  // namespace

extern "C" int dh_cellgene_graph_assemble(int64_t n_cells, int64_t n_genes, int64_t nnz, const int32_t* rowptr_x,
                                          const int32_t* col_x, const float* val_x, const int32_t* rowptr_t,
                                          const int32_t* col_t, const float* val_t, const int32_t* perm_t,
                                          int32_t* out_rowptr, int32_t* out_col, float* out_val, int32_t* out_eid,
                                          dh_stream_t stream) {
  if (n_cells < 0 || n_genes < 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "dh_cellgene_graph_assemble: negative size");
  if (2 * nnz + n_cells + n_genes >= ((int64_t)1 << 31))
    return dh::fail(DH_ERR_INVALID, "dh_cellgene_graph_assemble: edge count exceeds int32 CSR");
  if (!rowptr_x || !rowptr_t || !out_rowptr || !out_col || !out_val || !out_eid)
    return dh::fail(DH_ERR_INVALID, "dh_cellgene_graph_assemble: null pointer");
  const int64_t rows = n_cells + n_genes + 1;
  hipLaunchKernelGGL(cellgene_assemble_kernel, dim3((unsigned)dh::ceil_div(rows, 4)), dim3(256), 0, dh::as_stream(stream),
                     n_cells, n_genes, nnz, rowptr_x, col_x, val_x, rowptr_t, col_t, val_t, perm_t, out_rowptr, out_col,
                     out_val, out_eid);
  return dh::check_launch("dh_cellgene_graph_assemble");
}

// ---- dense -> CSR in row-major non-zero order: ``row, col = np.nonzero(feat)`` of cell_feature_graph.py:38 for an expression
// matrix that is already resident on the device (the on-device preprocessing pipeline).  One wavefront per row,
// 64 columns per step; the position of a non-zero inside its row is the popcount of the lower lanes' ballot bits, so the
// column order of np.nonzero is preserved exactly.
namespace {
__global__ __launch_bounds__(256) void dense_nnz_count_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                              int32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* x = X + row * ldx;
  int n = 0;
  for (int64_t c = lane; c < n_cols; c += 64) n += (x[c] != 0.f) ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
  if (lane == 0) counts[row] = n;
}

__global__ __launch_bounds__(256) void dense_to_csr_fill_kernel(int64_t n_rows, int64_t n_cols, const float* __restrict__ X, int64_t ldx,
                                                                const int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                                float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const float* x = X + row * ldx;
  int base = rowptr[row];
  const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int64_t c0 = 0; c0 < n_cols; c0 += 64) {
    const int64_t c = c0 + lane;
    const float v = (c < n_cols) ? x[c] : 0.f;
    const bool nz = v != 0.f;
    const uint64_t mask = __ballot(nz);
    if (nz) {
      const int pos = base + __popcll(mask & below);
      col[pos] = (int32_t)c;
      val[pos] = v;
    }
    base += __popcll(mask);
  }
}
}  // namespace

extern "C" int dh_dense_nnz_count_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, int32_t* counts, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_dense_nnz_count_f32: negative size");
  if (n_rows == 0) return DH_OK;
  if (!counts || (n_cols > 0 && !X)) return dh::fail(DH_ERR_INVALID, "dh_dense_nnz_count_f32: null pointer");
  if (ldx < n_cols) return dh::fail(DH_ERR_INVALID, "dh_dense_nnz_count_f32: leading dimension < n_cols");
  hipLaunchKernelGGL(dense_nnz_count_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X, ldx,
                     counts);
  return dh::check_launch("dh_dense_nnz_count_f32");
}

extern "C" int dh_dense_to_csr_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const int32_t* rowptr, int32_t* col,
                                   float* val, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0) return dh::fail(DH_ERR_INVALID, "dh_dense_to_csr_f32: negative size");
  if (n_rows == 0 || n_cols == 0) return DH_OK;
  if (!X || !rowptr || !col || !val) return dh::fail(DH_ERR_INVALID, "dh_dense_to_csr_f32: null pointer");
  if (ldx < n_cols) return dh::fail(DH_ERR_INVALID, "dh_dense_to_csr_f32: leading dimension < n_cols");
  hipLaunchKernelGGL(dense_to_csr_fill_kernel, dim3((unsigned)dh::ceil_div(n_rows, 4)), dim3(256), 0, dh::as_stream(stream), n_rows, n_cols, X,
                     ldx, rowptr, col, val);
  return dh::check_launch("dh_dense_to_csr_f32");
}
