// AdaptiveSAGE cell <- gene aggregation with the gene features STAGED IN LDS (SURVEY.md §8a A3, configs C3/C4;
// dance/models/nn/gnn.py:62-90).
//
//   neigh[v,:] = 1/deg(v) * ( sum_{gene edges e=(g->v)} w_e * (alpha[g] * H[g,:])  +  sum_{other edges} alpha[idx] w_e H[u,:] )
//
// Shape of the problem: ~2e8 edges point at only G ~ 2000 distinct gene rows (G*D*4 = 3.2 MB).  The generic SpMM
// (spmm.hip) re-gathers a 1600-byte row from L2 for every edge — 320 GB of L2 reads per pass at 1M cells, 13.5 ms at
// the 24 TB/s the L2 delivers.  Here a workgroup keeps a tile of the alpha-scaled gene features resident in its CU's
// 160 KB LDS for its whole life and streams cells past it:
//
//   * tile = GB genes x CW columns of fp32 (GB*CW*4 <= 160 000 B; D = 400 -> 4 column slices of 100, 5 gene blocks of 400);
//     bf16-stored features are widened once, while filling the tile, so the inner loop is the same fp32 loop;
//   * one wavefront owns one cell at a time; lane l owns columns [2l, 2l+1] of the slice (50 of 64 lanes at CW = 100).  The
//     cell's gene edges are loaded coalesced (lane l = edge l), staged in a 512-byte LDS strip of the wavefront and re-read at
//     a wave-uniform address (LDS broadcast); the gene row comes out of the tile as one conflict-free ds_read_b64 per lane and
//     the update is one v_pk_fma_f32;
//   * a cell's edges are sorted by gene id, so the part that falls in gene block b is a contiguous CSR segment; the
//     segment boundaries are found once per graph by a binary-search kernel (cacheable: reuse_segments);
//   * gene blocks are processed by successive launches: launch b continues the sequential fp32 sum of launch b-1 (partial
//     sums live in the output, or in an fp32 workspace when the output is bf16), so the summation order per element is
//     the CSR order — bit-reproducible, no atomics.  The last launch adds the non-gene edges (the cell's self loop) straight
//     from global memory, applies the mean, and stores;
//   * the workgroups that work on the same cells (one per column slice) are placed on the same XCD in the same round, so the
//     edge stream is fetched from HBM once and served to the other slices by that XCD's L2.
//
// MEASURED OUTCOME (round 2, 1M cells x 2000 genes, D = 400, 2.01e8 edges; scripts/sage_lds_bench.py): 13.8 ms fp32 and
// 13.5 ms bf16, against 13.5 / 7.8 ms of the L2-gather kernels — NOT faster, so the host keeps dispatching the gather kernels
// and this entry point stays an opt-in (kernels.sage_aggregate_cells) with its parity tests.  Why (ablation builds of this
// file, same inputs): walking the cells with the edge loop removed costs 5.1 ms — 20 (gene block x column slice) visits per
// cell, each only ~40 edges long, pay ~130 cycles per CU of per-visit work (bounds, edge loads, running-sum traffic); the
// running sums between gene blocks add 2.5 ms; the edge loop itself runs at 4 cycles per CU per (edge x 128 columns), 6.3 ms,
// where the LDS data path alone would allow 2.  Three ways of making the edge wave-uniform were measured and all land at
// 13.8-19 ms: scalar loads of the CSR stream (the scalar cache sustains only a handful of outstanding misses: streaming
// unique data through it is latency-bound, 17-19 ms), v_readlane (8 cycles each, two per edge: vector-ALU-bound, 14.6 ms;
// counters in profiles/r02_sage_lds_pmc.json), LDS broadcast (this file).  The tile cannot hold more than 400 genes x 100
// columns of fp32, which is what forces the 20 visits; see DESIGN.md for what would have to change (bf16 tile + MFMA).
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWaves = 16;              // 1024 threads: one workgroup per CU (LDS-limited), 4 waves per SIMD
constexpr int kLdsBytes = 154000;       // tile bytes; + kLdsPad + kStageBytes <= 160 KiB (163 840)
constexpr int kLdsPad = 512;            // a 64-lane ds_read_b64 of the last row may run this far past the tile
constexpr int kStageBytes = kWaves * 64 * 8;  // per wave: 64 staged edges (row index, weight) for uniform-address re-reads
constexpr int kGroup = 8;                // cells whose loads are in flight together, per wavefront
constexpr int kMaxBlocks = 64;          // gene blocks per call (G <= 64 * GB)

__device__ __forceinline__ float bf16_to_f32(uint16_t u) { return __uint_as_float((unsigned int)u << 16); }
__device__ __forceinline__ unsigned int f32_to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <typename T> __device__ __forceinline__ float load1(const T* p);
template <> __device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load1<uint16_t>(const uint16_t* p) { return bf16_to_f32(*p); }

template <typename T> __device__ __forceinline__ f32x2 load2(const T* p);
template <> __device__ __forceinline__ f32x2 load2<float>(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
template <> __device__ __forceinline__ f32x2 load2<uint16_t>(const uint16_t* p) {
  const unsigned int u = *reinterpret_cast<const unsigned int*>(p);
  f32x2 r;
  r.x = __uint_as_float(u << 16);
  r.y = __uint_as_float(u & 0xffff0000u);
  return r;
}
template <typename T> __device__ __forceinline__ void store2(T* p, f32x2 v);
template <> __device__ __forceinline__ void store2<float>(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
template <> __device__ __forceinline__ void store2<uint16_t>(uint16_t* p, f32x2 v) {
  *reinterpret_cast<unsigned int*>(p) = f32_to_bf16(v.x) | (f32_to_bf16(v.y) << 16);
}

struct Geometry {
  int n_slices, cw;     // column slices of cw columns (cw even, cw <= 128)
  int n_blocks, gb;     // gene blocks of gb genes
  int n_chunks;         // cell chunks (multiple of 8: chunk c runs on XCD c % 8)
};

// first position p in [lo, hi) whose key (col - gene_begin, unsigned) is >= x
__device__ __forceinline__ int lower_bound_key(const int32_t* __restrict__ col, int lo, int hi, unsigned int gene_begin, unsigned int x) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((unsigned int)col[mid] - gene_begin < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// seg[row][b] = first edge of `row` whose gene falls in block >= b (b = 0..n_blocks); seg[row][n_blocks] = first non-gene edge
__global__ __launch_bounds__(256) void sage_segments_kernel(int64_t n_dst, int n_blocks, int gb, int gene_begin, int gene_rows,
                                                            const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            int32_t* __restrict__ seg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_dst * (n_blocks + 1)) return;
  const int64_t row = i / (n_blocks + 1);
  const int b = (int)(i % (n_blocks + 1));
  const int s = rowptr[row], t = rowptr[row + 1];
  const long long x = (long long)b * gb;
  seg[i] = (b == 0) ? s : lower_bound_key(col, s, t, (unsigned int)gene_begin, (unsigned int)(x < gene_rows ? x : gene_rows));
}

struct Args {
  int64_t n_dst;
  int width, n_genes, gene_begin, gene_rows;
  int nnz;              // entries of col / w (scalar batches never read past it)
  const int32_t* rowptr;
  const int32_t* col;
  const float* w;
  const int32_t* src_id;
  const int32_t* dst_id;
  const float* alpha;
  const void* H;
  int64_t ldh;
  void* out;
  int64_t ldo;
  float* partial;       // fp32 running sums between gene blocks (== out when the output is fp32)
  int64_t ldp;
  const int32_t* seg;
};

__device__ __forceinline__ float sage_alpha_idx(const float* __restrict__ alpha, int n_genes, int sid, int did) {
  int idx = n_genes + 1;                      // cell self loop (default)
  if (sid >= 0 && did < 0) idx = sid;         // gene -> cell
  if (did >= 0 && sid < 0) idx = did;         // cell -> gene
  if (did >= 0 && sid >= 0) idx = n_genes;    // gene self loop
  return alpha[idx];
}

// The read-only streams are separate __restrict__ kernel arguments (not members of Args) so that alias analysis can prove
// the stores to partial / out never clobber them: only then does the compiler keep the wave-uniform loads on the scalar unit.
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kWaves * 64) void sage_cells_lds_kernel(Args a, Geometry g, int b, const int32_t* __restrict__ col,
                                                                       const float* __restrict__ wt, const int32_t* __restrict__ seg,
                                                                       const int32_t* __restrict__ rowptr,
                                                                       const int32_t* __restrict__ src_id,
                                                                       const int32_t* __restrict__ dst_id,
                                                                       const float* __restrict__ alpha) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [gb][cw]
  const f32x2* __restrict__ tile2 = reinterpret_cast<const f32x2*>(tile);
  const int cw2 = g.cw >> 1;
  // edge staging of this wavefront, behind the tile and its padding
  f32x2* stage = reinterpret_cast<f32x2*>(tile) + ((g.gb * g.cw * 4 + kLdsPad) >> 3) + (threadIdx.x >> 6) * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // blockIdx -> (chunk, slice): consecutive block ids go round the 8 XCDs; the n_slices workgroups of one chunk get ids
  // 8 apart, i.e. the same XCD and the same dispatch round
  const int j = blockIdx.x;
  const int xcd = j & 7, r = j >> 3;
  const int slice = r % g.n_slices;
  const int chunk = (r / g.n_slices) * 8 + xcd;
  const int col0 = slice * g.cw;
  const int cw = min(g.cw, a.width - col0);   // columns of this slice (even)
  const int g0 = b * g.gb;                     // first gene of this block (relative to gene_begin)
  const int ng = min(g.gb, a.gene_rows - g0);
  const TIN* __restrict__ H = static_cast<const TIN*>(a.H);

  // ---- fill: tile[q][c] = alpha[cell_id(gene)] * H[gene][col0 + c] ------------------------------------------------------
  for (int q = wave; q < ng; q += kWaves) {
    const int gene = a.gene_begin + g0 + q;
    const int sid = src_id[gene];
    const float al = alpha[sid >= 0 ? sid : a.n_genes + 1];  // gene -> cell factor (dst is a cell: did < 0)
    const TIN* hr = H + (int64_t)gene * a.ldh + col0;
    for (int c = lane; c < cw; c += 64) tile[q * g.cw + c] = al * load1<TIN>(hr + c);
  }
  __syncthreads();

  const int64_t per = (a.n_dst + g.n_chunks - 1) / g.n_chunks;
  const int64_t c_begin = (int64_t)chunk * per;
  const int64_t c_end = min(a.n_dst, c_begin + per);
  const bool first = (b == 0), last = (b == g.n_blocks - 1);
  const bool live = 2 * lane < cw;
  const int nb1 = g.n_blocks + 1;
  const int lane_off = lane * 2;
  const int gbase = a.gene_begin + g0;

  // Each wavefront owns a contiguous run of the chunk's cells.  Everything it needs per cell comes from HBM / the
  // Infinity Cache with ~2 us of latency (gene-edge segment, running sum), against ~0.25 us of work per cell, so the loads
  // are issued a whole group of kGroup cells ahead (2 x kGroup x 3 vector loads in flight per wavefront):
  //   per batch of 64 cells : lane l loads the segment bounds of cell l of the NEXT batch
  //   per group of 8 cells  : (row offset, weight) of every gene edge (lane l = edge l of the segment) and the running sum
  //                           of each cell of the NEXT group go into registers while the current group is processed
  //   per cell               : the staged edges are re-read from LDS at a wave-uniform address (broadcast), the gene row comes
  //                           out of the tile as one ds_read_b64 per lane, one packed FMA per edge
  const int64_t per_wave = (c_end - c_begin + kWaves - 1) / kWaves;
  const int64_t w_begin = c_begin + per_wave * wave;
  const int64_t w_end = min(c_end, w_begin + per_wave);
  auto load_bounds = [&](int64_t cb, int& s_out, int& t_out) {
    const int64_t cell = cb + lane;
    s_out = t_out = 0;
    if (cell < w_end) {
      s_out = seg[cell * nb1 + b];
      t_out = seg[cell * nb1 + b + 1];
    }
  };
  struct Group {
    int vc[kGroup];
    float vw[kGroup];
    f32x2 acc[kGroup];
  };
  // loads of cells [cb + i0, cb + i0 + kGroup) of the batch whose bounds sit in (sb, tb)
  auto issue = [&](Group& gr, int64_t cb, int i0, int n_here, int sb, int tb) {
#pragma unroll
    for (int u = 0; u < kGroup; ++u) {
      gr.vc[u] = 0;
      gr.vw[u] = 0.f;
      gr.acc[u] = f32x2{0.f, 0.f};
      if (i0 + u < n_here) {
        const int s = __builtin_amdgcn_readlane(sb, (i0 + u) & 63), t = __builtin_amdgcn_readlane(tb, (i0 + u) & 63);
        if (s + lane < t) {
          gr.vc[u] = (col[s + lane] - gbase) * cw2;
          gr.vw[u] = wt[s + lane];
        }
        if (!first && live) gr.acc[u] = *reinterpret_cast<const f32x2*>(a.partial + (cb + i0 + u) * a.ldp + col0 + lane_off);
      }
    }
  };
  int s0, t0, s1, t1;
  load_bounds(w_begin, s0, t0);
  Group cur, nxt;
  issue(nxt, w_begin, 0, (int)min((int64_t)64, w_end - w_begin), s0, t0);

  for (int64_t cb = w_begin; cb < w_end; cb += 64) {
    load_bounds(cb + 64, s1, t1);
    const int n_here = (int)min((int64_t)64, w_end - cb);
    const int n_next = (int)min((int64_t)64, w_end - (cb + 64));  // <= 0 when this is the last batch
    for (int i0 = 0; i0 < n_here; i0 += kGroup) {
      cur = nxt;
      if (i0 + kGroup < n_here) issue(nxt, cb, i0 + kGroup, n_here, s0, t0);
      else if (n_next > 0) issue(nxt, cb + 64, 0, n_next, s1, t1);
#pragma unroll
      for (int u = 0; u < kGroup; ++u) {
        const int i = i0 + u;
        if (i >= n_here) break;
        const int64_t cell = cb + i;
        const int s = __builtin_amdgcn_readlane(s0, i), t = __builtin_amdgcn_readlane(t0, i);
        f32x2 acc = cur.acc[u];
        int vc = cur.vc[u];
        float vw = cur.vw[u];
        for (int base = s; base < t; base += 64) {
          if (base > s) {  // segments longer than one wavefront of edges
            vc = 0;
            vw = 0.f;
            if (base + lane < t) {
              vc = (col[base + lane] - gbase) * cw2;
              vw = wt[base + lane];
            }
          }
          const int cnt = min(64, t - base);
          // stage the wavefront's edges in LDS; every lane then re-reads edge k at ONE address (an LDS broadcast), which costs
          // no vector-ALU issue slot (v_readlane measured ~8 cycles each: with two of them per edge the loop was ALU-bound)
          {
            f32x2 pr;
            pr.x = __int_as_float(vc);
            pr.y = vw;
            stage[lane] = pr;
          }
          int k = 0;
          for (; k + 8 <= cnt; k += 8) {
            f32x2 ed[8], z[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) ed[v] = stage[k + v];
#pragma unroll
            for (int v = 0; v < 8; ++v) z[v] = tile2[__float_as_int(ed[v].x) + lane];  // lanes past the slice read on into the next row
#pragma unroll
            for (int v = 0; v < 8; ++v) {
              acc.x = fmaf(ed[v].y, z[v].x, acc.x);
              acc.y = fmaf(ed[v].y, z[v].y, acc.y);
            }
          }
          for (; k < cnt; ++k) {
            const f32x2 ed = stage[k];
            const f32x2 z = tile2[__float_as_int(ed.x) + lane];
            acc.x = fmaf(ed.y, z.x, acc.x);
            acc.y = fmaf(ed.y, z.y, acc.y);
          }
        }
        if (!last) {
          if (live) *reinterpret_cast<f32x2*>(a.partial + cell * a.ldp + col0 + lane_off) = acc;
          continue;
        }
        // last gene block: the remaining (non-gene) edges straight from global memory, then the mean
        const int rs = rowptr[cell], rt = rowptr[cell + 1];
        const int did = dst_id[cell];
        for (int k = t; k < rt; ++k) {
          const int c = col[k];
          const float f = wt[k] * sage_alpha_idx(alpha, a.n_genes, src_id[c], did);
          if (live) {
            const f32x2 z = load2<TIN>(H + (int64_t)c * a.ldh + col0 + lane_off);
            acc.x = fmaf(f, z.x, acc.x);
            acc.y = fmaf(f, z.y, acc.y);
          }
        }
        const float scale = (rt > rs) ? 1.f / (float)(rt - rs) : 0.f;
        acc.x *= scale;
        acc.y *= scale;
        if (live) store2<TOUT>(static_cast<TOUT*>(a.out) + cell * a.ldo + col0 + lane_off, acc);
      }
    }
    s0 = s1;
    t0 = t1;
  }
}

Geometry make_geometry(int64_t n_dst, int width, int gene_rows) {
  Geometry g;
  g.n_slices = (width + 127) / 128;
  g.cw = (((width + g.n_slices - 1) / g.n_slices) + 1) & ~1;
  const int max_gb = kLdsBytes / (g.cw * 4);
  g.n_blocks = (gene_rows + max_gb - 1) / max_gb;
  if (g.n_blocks < 1) g.n_blocks = 1;
  g.gb = (gene_rows + g.n_blocks - 1) / g.n_blocks;
  // one workgroup per CU: whole rounds of the 256 CUs when there is enough work, at least ~64 cells per wave otherwise
  int64_t chunks = 256 / g.n_slices;
  if (chunks < 8) chunks = 8;
  const int64_t min_cells = (int64_t)kWaves * 16;
  while (chunks > 8 && n_dst / chunks < min_cells) chunks -= 8;
  g.n_chunks = (int)((chunks + 7) / 8 * 8);
  return g;
}

template <typename TIN, typename TOUT>
int launch(const Args& a, const Geometry& g, hipStream_t st) {
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sage_cells_lds_kernel<TIN, TOUT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes + kLdsPad + kStageBytes) == hipSuccess;
  if (!ok) return dh::fail(DH_ERR_LAUNCH, "dh_sage_aggregate_cells: cannot reserve %d bytes of LDS", kLdsBytes);
  const size_t lds = (size_t)g.gb * g.cw * 4 + kLdsPad + kStageBytes;
  for (int b = 0; b < g.n_blocks; ++b)
    hipLaunchKernelGGL((sage_cells_lds_kernel<TIN, TOUT>), dim3((unsigned)(g.n_chunks * g.n_slices)), dim3(kWaves * 64), lds, st, a, g, b,
                       a.col, a.w, a.seg, a.rowptr, a.src_id, a.dst_id, a.alpha);
  return dh::check_launch("dh_sage_aggregate_cells");
}

}  // namespace

extern "C" size_t dh_sage_cells_workspace_bytes(int64_t n_dst, int64_t width, int64_t gene_rows, int out_dtype) {
  if (n_dst <= 0 || width <= 0 || gene_rows <= 0) return 0;
  const Geometry g = make_geometry(n_dst, (int)width, (int)gene_rows);
  size_t bytes = (size_t)n_dst * (g.n_blocks + 1) * sizeof(int32_t);          // segment table
  bytes = (bytes + 255) & ~(size_t)255;
  if (out_dtype == DH_DTYPE_BF16 && g.n_blocks > 1) bytes += (size_t)n_dst * width * sizeof(float);  // fp32 running sums
  return bytes;
}

extern "C" int dh_sage_aggregate_cells(int64_t n_dst, int64_t n_src, int64_t nnz, int64_t width, int64_t n_genes, int64_t gene_begin,
                                       int64_t gene_rows, const int32_t* rowptr, const int32_t* col, const float* w,
                                       const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha,
                                       const void* H, int64_t ldh, int h_dtype, void* neigh, int64_t ldn, int out_dtype,
                                       void* workspace, size_t workspace_bytes, int reuse_segments, dh_stream_t stream) {
  const char* me = "dh_sage_aggregate_cells";
  if (n_dst < 0 || n_src < 0 || nnz < 0 || nnz >= ((int64_t)1 << 31) || width < 0 || n_genes < 0 || gene_begin < 0 || gene_rows < 0)
    return dh::fail(DH_ERR_INVALID, "%s: negative size (or nnz >= 2^31)", me);
  if (n_dst == 0 || width == 0) return DH_OK;
  if (!rowptr || !col || !w || !src_cell_id || !dst_cell_id || !alpha || !H || !neigh) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (gene_rows == 0 || gene_begin + gene_rows > n_src) return dh::fail(DH_ERR_INVALID, "%s: gene rows [%lld, %lld) outside the %lld source rows", me,
                                                                       (long long)gene_begin, (long long)(gene_begin + gene_rows), (long long)n_src);
  if ((h_dtype != DH_DTYPE_F32 && h_dtype != DH_DTYPE_BF16) || (out_dtype != DH_DTYPE_F32 && out_dtype != DH_DTYPE_BF16))
    return dh::fail(DH_ERR_INVALID, "%s: bad dtype", me);
  if (width % 2 || ldh % 2 || ldn % 2 || ldh < width || ldn < width) return dh::fail(DH_ERR_INVALID, "%s: needs even width / leading dimensions >= width", me);
  const size_t in_align = h_dtype == DH_DTYPE_F32 ? 8 : 4, out_align = out_dtype == DH_DTYPE_F32 ? 8 : 4;
  if ((uintptr_t)H % in_align || (uintptr_t)neigh % out_align) return dh::fail(DH_ERR_INVALID, "%s: misaligned H / neigh", me);
  if (n_dst >= ((int64_t)1 << 31) / 80 || n_src >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: too many rows", me);
  const Geometry g = make_geometry(n_dst, (int)width, (int)gene_rows);
  if (g.n_blocks > kMaxBlocks) return dh::fail(DH_ERR_INVALID, "%s: %lld gene rows need more than %d gene blocks", me, (long long)gene_rows, kMaxBlocks);
  if (workspace_bytes < dh_sage_cells_workspace_bytes(n_dst, width, gene_rows, out_dtype) || !workspace)
    return dh::fail(DH_ERR_INVALID, "%s: workspace too small (see dh_sage_cells_workspace_bytes)", me);
  hipStream_t st = dh::as_stream(stream);
  int32_t* seg = static_cast<int32_t*>(workspace);
  size_t seg_bytes = ((size_t)n_dst * (g.n_blocks + 1) * sizeof(int32_t) + 255) & ~(size_t)255;
  if (!reuse_segments) {
    const int64_t total = n_dst * (g.n_blocks + 1);
    hipLaunchKernelGGL(sage_segments_kernel, dim3((unsigned)dh::ceil_div(total, 256)), dim3(256), 0, st, n_dst, g.n_blocks, g.gb,
                       (int)gene_begin, (int)gene_rows, rowptr, col, seg);
  }
  Args a;
  a.n_dst = n_dst; a.width = (int)width; a.n_genes = (int)n_genes; a.gene_begin = (int)gene_begin; a.gene_rows = (int)gene_rows;
  a.nnz = (int)nnz;
  a.rowptr = rowptr; a.col = col; a.w = w; a.src_id = src_cell_id; a.dst_id = dst_cell_id; a.alpha = alpha;
  a.H = H; a.ldh = ldh; a.out = neigh; a.ldo = ldn; a.seg = seg;
  if (out_dtype == DH_DTYPE_F32) { a.partial = static_cast<float*>(neigh); a.ldp = ldn; }
  else { a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + seg_bytes); a.ldp = width; }
  if (h_dtype == DH_DTYPE_F32 && out_dtype == DH_DTYPE_F32) return launch<float, float>(a, g, st);
  if (h_dtype == DH_DTYPE_BF16 && out_dtype == DH_DTYPE_BF16) return launch<uint16_t, uint16_t>(a, g, st);
  if (h_dtype == DH_DTYPE_BF16 && out_dtype == DH_DTYPE_F32) return launch<uint16_t, float>(a, g, st);
  return launch<float, uint16_t>(a, g, st);
}
