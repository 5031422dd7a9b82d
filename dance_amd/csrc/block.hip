// Device-side message-flow block builder: the full-fan-out in-neighbour blocks that
// dgl.dataloading.NeighborSampler([-1] * L, edge_dir="in") / MultiLayerFullNeighborSampler hand to scDeepSort and
// graph-sc (dance/modules/single_modality/cell_type_annotation/scdeepsort.py:183,233-236; clustering/graphsc.py:181-183).
//
// A block for `seeds` holds every in-edge of the seeds.  Its source nodes are the seeds first (dgl.to_block puts the
// destination nodes first) followed by the remaining in-neighbours in ascending node id; block columns are positions in that
// source list.  The torch-op version (a dozen index kernels and two host round trips per batch) was half of a scDeepSort
// epoch at 1M cells; here a block costs two C calls with ONE host read of two integers between them:
//
//   dh_block_plan : degrees of the seed rows -> block row pointers (scan); byte-mark every in-neighbour; un-mark the seeds and
//                   record their positions in the node -> block-position table `lut`; count the marks per 2048-node chunk
//                   (scan) -> totals {number of block edges, number of non-seed sources}
//   dh_block_fill : ordered compaction of the marks into the source list (ascending id), positions into `lut`, marks
//                   cleared again (the mark array is all-zero between calls: no O(N) memset per batch); block columns =
//                   lut[graph column], values copied — one wavefront per seed row, coalesced.
//
// `mark` (uint8[n_nodes], zero-initialised once) and `lut` (int32[n_nodes], never initialised) persist on the graph.  Seeds
// must be unique (they are a batch of a permutation in every caller).
#include <algorithm>

#include "common.h"

extern "C" size_t dh_exclusive_scan_i32_workspace_bytes(int64_t n);
extern "C" int dh_exclusive_scan_i32(int64_t n, const int32_t* in, int32_t* out, void* workspace, size_t workspace_bytes, dh_stream_t stream);

namespace {

constexpr int kChunk = 2048;  // nodes per compaction chunk (256 threads x 8 bytes)

__global__ __launch_bounds__(256) void block_deg_kernel(int64_t n_seeds, const int64_t* __restrict__ seeds, const int32_t* __restrict__ rowptr,
                                                        int32_t* __restrict__ deg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_seeds) return;
  const int64_t v = seeds[i];
  deg[i] = rowptr[v + 1] - rowptr[v];
}

// one wavefront per seed row
__global__ __launch_bounds__(256) void block_mark_kernel(int64_t n_seeds, const int64_t* __restrict__ seeds, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, uint8_t* __restrict__ mark) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_seeds) return;
  const int64_t v = seeds[i];
  const int s = rowptr[v], t = rowptr[v + 1];
  for (int e = s + lane; e < t; e += 64) mark[col[e]] = 1;
}

__global__ __launch_bounds__(256) void block_seed_kernel(int64_t n_seeds, const int64_t* __restrict__ seeds, uint8_t* __restrict__ mark,
                                                         int32_t* __restrict__ lut) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_seeds) return;
  const int64_t v = seeds[i];
  mark[v] = 0;
  lut[v] = (int32_t)i;
}

__device__ __forceinline__ int popcount8(const uint8_t* __restrict__ mark, int64_t base, int64_t n_nodes, unsigned long long* bits) {
  // 8 consecutive marks of this thread as a bit field
  unsigned long long m = 0;
  if (base + 8 <= n_nodes) {
    const unsigned long long w = *reinterpret_cast<const unsigned long long*>(mark + base);  // bytes are 0 / 1
    m = w;
  } else {
    for (int k = 0; k < 8; ++k)
      if (base + k < n_nodes) m |= (unsigned long long)mark[base + k] << (8 * k);
  }
  *bits = m;
  return __popcll(m);
}

__global__ __launch_bounds__(256) void block_count_kernel(int64_t n_nodes, const uint8_t* __restrict__ mark, int32_t* __restrict__ counts) {
  __shared__ int part[4];
  const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * 8;
  unsigned long long bits;
  int c = base < n_nodes ? popcount8(mark, base, n_nodes, &bits) : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void block_totals_kernel(const int32_t* __restrict__ brp, int64_t n_seeds, const int32_t* __restrict__ cbase, int64_t n_chunks,
                                    int32_t* __restrict__ totals) {
  totals[0] = brp[n_seeds];
  totals[1] = cbase[n_chunks];
}

// ordered compaction: thread t of chunk c owns nodes [c * 2048 + 8 t, + 8)
__global__ __launch_bounds__(256) void block_others_kernel(int64_t n_nodes, int64_t n_seeds, uint8_t* __restrict__ mark,
                                                           const int32_t* __restrict__ cbase, int32_t* __restrict__ lut,
                                                           int64_t* __restrict__ src_ids) {
  __shared__ int wave_tot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * 8;
  unsigned long long bits = 0;
  const int c = base < n_nodes ? popcount8(mark, base, n_nodes, &bits) : 0;
  int incl = c;  // inclusive scan inside the wavefront
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int before = cbase[blockIdx.x];
  for (int w2 = 0; w2 < wave; ++w2) before += wave_tot[w2];
  int rank = before + incl - c;
  if (c == 0) return;
  for (int k = 0; k < 8; ++k) {
    if ((bits >> (8 * k)) & 1ull) {
      const int64_t node = base + k;
      src_ids[n_seeds + rank] = node;
      lut[node] = (int32_t)(n_seeds + rank);
      mark[node] = 0;  // leave the array all-zero for the next block
      ++rank;
    }
  }
}

__global__ __launch_bounds__(256) void block_fill_kernel(int64_t n_seeds, const int64_t* __restrict__ seeds, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, const float* __restrict__ val,
                                                         const int32_t* __restrict__ brp, const int32_t* __restrict__ lut,
                                                         int32_t* __restrict__ bcol, float* __restrict__ bval, int64_t* __restrict__ src_ids) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_seeds) return;
  const int64_t v = seeds[i];
  if (lane == 0) src_ids[i] = v;
  const int s = rowptr[v], t = rowptr[v + 1], o = brp[i];
  for (int e = lane; e < t - s; e += 64) {
    bcol[o + e] = lut[col[s + e]];
    if (val) bval[o + e] = val[s + e];
  }
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Layout {
  size_t deg, counts, cbase, scan, total;
  int64_t n_chunks;
};
Layout make_layout(int64_t n_nodes, int64_t n_seeds) {
  Layout l;
  l.n_chunks = dh::ceil_div(n_nodes, kChunk);
  l.deg = 0;
  l.counts = l.deg + align256((size_t)n_seeds * 4);
  l.cbase = l.counts + align256((size_t)l.n_chunks * 4);
  l.scan = l.cbase + align256((size_t)(l.n_chunks + 1) * 4);
  const size_t s1 = dh_exclusive_scan_i32_workspace_bytes(n_seeds), s2 = dh_exclusive_scan_i32_workspace_bytes(l.n_chunks);
  l.total = l.scan + align256(s1 > s2 ? s1 : s2);
  return l;
}

}  // namespace

extern "C" size_t dh_block_workspace_bytes(int64_t n_nodes, int64_t n_seeds) {
  if (n_nodes <= 0 || n_seeds <= 0) return 0;
  return make_layout(n_nodes, n_seeds).total;
}

extern "C" int dh_block_plan(int64_t n_nodes, int64_t n_seeds, const int64_t* seeds, const int32_t* rowptr, const int32_t* col,
                             uint8_t* mark, int32_t* lut, int32_t* block_rowptr, int32_t* totals, void* workspace,
                             size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_block_plan";
  if (n_nodes < 0 || n_seeds < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (!block_rowptr || !totals) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  hipStream_t st = dh::as_stream(stream);
  if (n_seeds == 0 || n_nodes == 0) {
    if (dh::zero_async(block_rowptr, sizeof(int32_t), st) != hipSuccess || dh::zero_async(totals, 2 * sizeof(int32_t), st) != hipSuccess)
      return dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me);
    return DH_OK;
  }
  if (!seeds || !rowptr || !col || !mark || !lut) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (n_nodes >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: n_nodes >= 2^31", me);
  const Layout l = make_layout(n_nodes, n_seeds);
  if (!workspace || workspace_bytes < l.total) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, workspace_bytes, l.total);
  char* ws = static_cast<char*>(workspace);
  int32_t* deg = reinterpret_cast<int32_t*>(ws + l.deg);
  int32_t* counts = reinterpret_cast<int32_t*>(ws + l.counts);
  int32_t* cbase = reinterpret_cast<int32_t*>(ws + l.cbase);
  hipLaunchKernelGGL(block_deg_kernel, dim3((unsigned)dh::ceil_div(n_seeds, 256)), dim3(256), 0, st, n_seeds, seeds, rowptr, deg);
  int rc = dh_exclusive_scan_i32(n_seeds, deg, block_rowptr, ws + l.scan, l.total - l.scan, stream);
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(block_mark_kernel, dim3((unsigned)dh::ceil_div(n_seeds, 4)), dim3(256), 0, st, n_seeds, seeds, rowptr, col, mark);
  hipLaunchKernelGGL(block_seed_kernel, dim3((unsigned)dh::ceil_div(n_seeds, 256)), dim3(256), 0, st, n_seeds, seeds, mark, lut);
  hipLaunchKernelGGL(block_count_kernel, dim3((unsigned)l.n_chunks), dim3(256), 0, st, n_nodes, mark, counts);
  rc = dh_exclusive_scan_i32(l.n_chunks, counts, cbase, ws + l.scan, l.total - l.scan, stream);
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(block_totals_kernel, dim3(1), dim3(1), 0, st, block_rowptr, n_seeds, cbase, l.n_chunks, totals);
  return dh::check_launch(me);
}

extern "C" int dh_block_fill(int64_t n_nodes, int64_t n_seeds, const int64_t* seeds, const int32_t* rowptr, const int32_t* col,
                             const float* val, uint8_t* mark, int32_t* lut, const int32_t* block_rowptr, int32_t* block_col,
                             float* block_val, int64_t* src_ids, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_block_fill";
  if (n_nodes < 0 || n_seeds < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_seeds == 0 || n_nodes == 0) return DH_OK;
  if (!seeds || !rowptr || !col || !mark || !lut || !block_rowptr || !src_ids) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  const Layout l = make_layout(n_nodes, n_seeds);
  if (!workspace || workspace_bytes < l.total) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes (the workspace of dh_block_plan)", me, workspace_bytes, l.total);
  hipStream_t st = dh::as_stream(stream);
  const int32_t* cbase = reinterpret_cast<const int32_t*>(static_cast<char*>(workspace) + l.cbase);
  hipLaunchKernelGGL(block_others_kernel, dim3((unsigned)l.n_chunks), dim3(256), 0, st, n_nodes, n_seeds, mark, cbase, lut, src_ids);
  hipLaunchKernelGGL(block_fill_kernel, dim3((unsigned)dh::ceil_div(n_seeds, 4)), dim3(256), 0, st, n_seeds, seeds, rowptr, col, val, block_rowptr, lut,
                     block_col, block_val, src_ids);
  return dh::check_launch(me);
}

// ---- static-shape block of seed CELLS of a CellFeatureGraph-layout graph (genes are nodes [0, G), every cell's in-edges are its
// genes, ascending, then its own self loop) ---------------------------------------------------------------------------------------
// Source list = [the B seeds | ALL G genes]: a superset of dgl.to_block's (genes no seed expresses get no edge, so every layer
// output is unchanged), but of a size that does not depend on the batch — the block needs no marking / compaction, and no value
// ever has to reach the host: every buffer has a static shape, so a whole training step on it can be captured as ONE hipGraph
// (graph-sc / scDeepSort at the reference's batch sizes are launch-bound otherwise).
//   rows 0 .. B-1 : the seeds' in-edges, column = B + gene id, resp. the row's own index for the self loop
//   row  B        : a padding row owning the unused tail [nnz, E_max) of the edge arrays (column 0, value 0): the CSR always has
//                   exactly E_max entries and B + 1 rows, its transpose is well defined, the padding contributes zeros.
// bad[0] collects flags, checked by the caller once per epoch: bit 1 = a seed is not a cell of that layout (a non-gene in-neighbour
// other than the seed itself) or the block needs more than E_max edges; bit 2 = a seed without exactly one self loop (only a caller
// that relies on the identity among a batch's own cells — graph-sc's decoder target — has to care).
namespace {

__global__ __launch_bounds__(256) void cells_static_fill_kernel(int64_t n_seeds, int64_t n_genes, int64_t e_max, const int64_t* __restrict__ seeds,
                                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ val, int32_t* __restrict__ brp,
                                                                int32_t* __restrict__ bcol, float* __restrict__ bval, int32_t* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i > n_seeds) return;
  const int nnz = brp[n_seeds];
  if (i == n_seeds) {  // the padding row: one wavefront is enough for its bookkeeping, the tail itself is cleared below
    if (lane == 0) {
      brp[n_seeds + 1] = (int32_t)e_max;
      if (nnz > e_max) atomicOr(bad, 1);
    }
    return;
  }
  const int64_t v = seeds[i];
  const int s = rowptr[v], t = rowptr[v + 1], o = brp[i];
  if (o < 0 || o + (t - s) > e_max) {  // too many entries: flagged by the padding row's wavefront; a NEGATIVE offset can only come from
    if (o < 0 && lane == 0) atomicOr(bad, 1);  // corrupted row pointers (the replayed memset node of round 5's hunt) — never store below the buffer
    return;
  }
  int n_self = 0;  // complete in lane 0 (it is active in every trip any lane makes)
  for (int e = lane; e < t - s; e += 64) {
    const int c = col[s + e];
    int bc;
    if (c < n_genes) bc = (int)n_seeds + c;
    else {
      bc = (int)i;
      if (c != v) atomicOr(bad, 1);  // a cell -> cell edge other than the self loop: not a CellFeatureGraph row
    }
    n_self += __popcll(__ballot(c >= n_genes));
    bcol[o + e] = bc;
    bval[o + e] = val ? val[s + e] : 1.f;
  }
  // graph-sc's captured step takes the decoder target among a batch's own cells to be the identity (graphsc.py:208-214 on a
  // CellFeatureGraph: the only cell -> cell edges are the self loops): a seed row without exactly one self loop breaks THAT, and only
  // that — its own bit (2), which ScDeepSort's captured step (no decoder, no identity assumption) does not look at
  if (lane == 0 && n_self != 1) atomicOr(bad, 2);
}

__global__ __launch_bounds__(256) void cells_static_pad_kernel(int64_t n_seeds, int64_t e_max, const int32_t* __restrict__ brp,
                                                               int32_t* __restrict__ bcol, float* __restrict__ bval) {
  const int64_t nnz = brp[n_seeds];
  for (int64_t e = nnz + (int64_t)blockIdx.x * 256 + threadIdx.x; e < e_max; e += (int64_t)gridDim.x * 256) {
    bcol[e] = 0;
    bval[e] = 0.f;
  }
}

// ---- degree scalings of a block (dgl GraphConv norm="both" / "right"; graphsc.py:444-474) -----------------------------------------------
// rowscale[i] = f(in-degree of row i), colscale[j] = max(out-degree of column j, 1)^-1/2 counted over the entries of rows [0, n_rows) only
// (a static block's padding tail lies behind them).  One memset + two launches; the torch formulation was 14 (arange / compare / casts /
// zeros / index_add / clamp / rsqrt ...), i.e. 14 of the ~100 five-microsecond kernels of a captured batch-128 step.
__global__ __launch_bounds__(256) void degree_count_kernel(int64_t n_rows, int64_t n_pad, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           int mode, float* __restrict__ rowscale, int32_t* __restrict__ count) {
  const int64_t nnz = rowptr[n_rows];
  if (blockIdx.x == 0 && threadIdx.x < n_pad) rowscale[n_rows + threadIdx.x] = 1.f;  // padding rows scale by one
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) {
    if (count) atomicAdd(count + col[e], 1);
    if (e < n_rows) {
      const float d = fmaxf((float)(rowptr[e + 1] - rowptr[e]), 1.f);
      rowscale[e] = mode == DH_DEGREE_BOTH ? 1.f / sqrtf(d) : 1.f / d;
    }
  }
  // (rows outnumber entries only for blocks without edges: finish the row scales)
  for (int64_t i = max(nnz, (int64_t)0) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * 256) {
    const float d = fmaxf((float)(rowptr[i + 1] - rowptr[i]), 1.f);
    rowscale[i] = mode == DH_DEGREE_BOTH ? 1.f / sqrtf(d) : 1.f / d;
  }
}

__global__ __launch_bounds__(256) void degree_colscale_kernel(int64_t n_cols, const int32_t* __restrict__ count, float* __restrict__ colscale) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n_cols) colscale[j] = 1.f / sqrtf(fmaxf((float)count[j], 1.f));  // correctly rounded sqrt and divide (within one rounding of pow(d, -0.5))
}

}  // namespace

extern "C" size_t dh_block_cells_static_workspace_bytes(int64_t n_seeds) {
  if (n_seeds <= 0) return 0;
  return align256((size_t)n_seeds * 4) + align256(dh_exclusive_scan_i32_workspace_bytes(n_seeds));
}

extern "C" int dh_block_cells_static(int64_t n_seeds, int64_t n_genes, int64_t e_max, const int64_t* seeds, const int32_t* rowptr,
                                     const int32_t* col, const float* val, int32_t* block_rowptr, int32_t* block_col, float* block_val,
                                     int32_t* bad, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_block_cells_static";
  if (n_seeds <= 0 || n_genes < 0 || e_max < 0) return dh::fail(DH_ERR_INVALID, "%s: bad size", me);
  if (!seeds || !rowptr || !col || !block_rowptr || !block_col || !block_val || !bad) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  const size_t deg_bytes = align256((size_t)n_seeds * 4);
  if (!workspace || workspace_bytes < dh_block_cells_static_workspace_bytes(n_seeds))
    return dh::fail(DH_ERR_WORKSPACE, "%s: workspace too small", me);
  hipStream_t st = dh::as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int32_t* deg = reinterpret_cast<int32_t*>(ws);
  hipLaunchKernelGGL(block_deg_kernel, dim3((unsigned)dh::ceil_div(n_seeds, 256)), dim3(256), 0, st, n_seeds, seeds, rowptr, deg);
  const int rc = dh_exclusive_scan_i32(n_seeds, deg, block_rowptr, ws + deg_bytes, workspace_bytes - deg_bytes, stream);  // block_rowptr[0 .. B]
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(cells_static_fill_kernel, dim3((unsigned)dh::ceil_div(n_seeds + 1, 4)), dim3(256), 0, st, n_seeds, n_genes, e_max, seeds, rowptr,
                     col, val, block_rowptr, block_col, block_val, bad);
  const unsigned pgrid = (unsigned)std::min<int64_t>(dh::ceil_div(e_max, 256), 1024);
  if (pgrid > 0)
    hipLaunchKernelGGL(cells_static_pad_kernel, dim3(pgrid), dim3(256), 0, st, n_seeds, e_max, block_rowptr, block_col, block_val);
  return dh::check_launch(me);
}

extern "C" int dh_csr_degree_scales_f32(int64_t n_rows, int64_t n_pad, int64_t n_cols, const int32_t* rowptr, const int32_t* col, int mode,
                                        float* rowscale, float* colscale, int32_t* count, dh_stream_t stream) {
  const char* me = "dh_csr_degree_scales_f32";
  if (n_rows < 0 || n_cols < 0 || n_pad < 0 || n_pad > 256) return dh::fail(DH_ERR_INVALID, "%s: bad size (n_pad <= 256)", me);
  if (mode != DH_DEGREE_BOTH && mode != DH_DEGREE_MEAN) return dh::fail(DH_ERR_INVALID, "%s: bad mode %d", me, mode);
  if (n_rows == 0 && n_cols == 0) return DH_OK;
  if (!rowptr || (n_rows > 0 && !rowscale)) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  const bool cols = mode == DH_DEGREE_BOTH;
  if (cols && n_cols > 0 && (!colscale || !count)) return dh::fail(DH_ERR_INVALID, "%s: mode BOTH needs colscale and the count buffer [n_cols]", me);  // (col may be null for a block without entries)
  hipStream_t st = dh::as_stream(stream);
  if (cols && n_cols > 0) (void)dh::zero_async(count, (size_t)n_cols * sizeof(int32_t), st);
  hipLaunchKernelGGL(degree_count_kernel, dim3(1024), dim3(256), 0, st, n_rows, n_pad, rowptr, col, mode, rowscale, cols && n_cols > 0 ? count : nullptr);
  int rc = dh::check_launch(me);
  if (rc != DH_OK || !cols || n_cols == 0) return rc;
  hipLaunchKernelGGL(degree_colscale_kernel, dim3((unsigned)dh::ceil_div(n_cols, 256)), dim3(256), 0, st, n_cols, count, colscale);
  return dh::check_launch(me);
}
