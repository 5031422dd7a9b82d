// Exact kNN of LOW-DIMENSIONAL points (d <= 3: the spatial coordinates of SpaGCN / STAGATE graphs, dance/transforms/graph/
// spatial_graph.py:143-151 NearestNeighbors on `spatial`; the 2-d / 3-d layouts neighbor_graph.py is run on) by a uniform cell grid —
// the DH_KNN_GRID algorithm of dh_knn_bruteforce_f32, same contract as the scan (knn.hip): d2 of a pair is the f32 chain
// ((0 + sq(x_0 - y_0)) + sq(x_1 - y_1)) + ..., the neighbours are the k smallest (d2, index) pairs, ties to the lower index, the query
// itself included; distances leave as correctly rounded square roots.  Bit-identical to the scan by construction: the same chain on
// every examined pair, and a pair is only left unexamined when its distance provably exceeds the current k-th.
//
// Why: all-pairs work on 500k three-dimensional points is 2.5e11 pair evaluations whatever engine runs them — the matrix-core filter
// + re-rank took 0.30 s for BASELINE config 5's graph (5 x 60 ms knn_rerank: profiles/r05x_c5_spagcn_kernel_stats_final.md) next to a
// 0.7 ms training iteration.  In 3 dimensions the k nearest points live in the 27 cells around the query: ~100 pair evaluations each.
//
//   grid_init   : bounding box registers, cell counters
//   grid_bbox   : min / max per dimension (ordered-integer atomics)
//   grid_params : ONE thread: cell edge h by bisection so that the grid has ~n / kTargetPerCell cells (<= n), dims, slack — on the
//                 device, so the launcher never reads anything back
//   grid_count  : cell of every point, population per cell                (integer atomics)
//   scan        : cell start offsets (dh_exclusive_scan_i32)
//   grid_scatter: points into cell order as (x, y, z, index) quadruples   (order inside a cell is arbitrary: the result does not depend on it)
//   grid_query  : one thread per point, in cell order (neighbouring lanes walk neighbouring cells): rings of cells of growing Chebyshev
//                 radius r around the query's cell; after ring r every unexamined point is at least `bound` away — the distance to the
//                 nearest face of the examined block that is not also the grid's own boundary — and the search stops once the k-th best
//                 d2 is below (bound - slack)^2; slack covers the rounding of the cell assignment and of the face positions.  Past ring
//                 kMaxRing (an outlier far from everything) the query scans all points — exact either way.
#include <algorithm>

#include "common.h"

extern "C" size_t dh_exclusive_scan_i32_workspace_bytes(int64_t n);
extern "C" int dh_exclusive_scan_i32(int64_t n, const int32_t* in, int32_t* out, void* workspace, size_t workspace_bytes, dh_stream_t stream);

namespace {

constexpr int kTargetPerCell = 4;
constexpr int kMaxRing = 6;
constexpr int QB = 256;

struct GridParams {
  float lo[3], h, inv_h, slack;
  int dim[3];
  int n_cells;
};

__device__ __forceinline__ unsigned enc(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ __launch_bounds__(256) void grid_init_kernel(int64_t n_cells_max, unsigned* __restrict__ bbox, int32_t* __restrict__ count, int32_t* __restrict__ cursor) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < 3) {
    bbox[i] = 0xFFFFFFFFu;  // running minimum
    bbox[3 + i] = 0u;       // running maximum
  }
  for (int64_t j = i; j < n_cells_max; j += (int64_t)gridDim.x * 256) {
    count[j] = 0;
    cursor[j] = 0;
  }
}

__global__ __launch_bounds__(256) void grid_bbox_kernel(int64_t n, int d, const float* __restrict__ X, int64_t ldx, unsigned* __restrict__ bbox) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    for (int j = 0; j < d; ++j) {
      const float v = X[i * ldx + j];
      lo[j] = fminf(lo[j], v);
      hi[j] = fmaxf(hi[j], v);
    }
  for (int j = 0; j < d; ++j) {
    float a = lo[j], b = hi[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = fminf(a, __shfl_xor(a, off, 64));
      b = fmaxf(b, __shfl_xor(b, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && a <= b) {
      atomicMin(bbox + j, enc(a));
      atomicMax(bbox + 3 + j, enc(b));
    }
  }
}

__device__ double cells_at(const float* ext, int d, double h) {
  double c = 1.0;
  for (int j = 0; j < d; ++j) c *= fmax(1.0, ceil((double)ext[j] / h));
  return c;
}

__global__ void grid_params_kernel(int64_t n, int d, int64_t n_cells_max, const unsigned* __restrict__ bbox, GridParams* __restrict__ P) {
  float ext[3] = {0.f, 0.f, 0.f}, maxabs = 0.f;
  for (int j = 0; j < 3; ++j) {
    const float a = j < d ? dec(bbox[j]) : 0.f, b = j < d ? dec(bbox[3 + j]) : 0.f;
    P->lo[j] = a;
    ext[j] = j < d ? fmaxf(b - a, 0.f) : 0.f;
    maxabs = fmaxf(maxabs, fmaxf(fabsf(a), fabsf(b)));
  }
  float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
  const double want = fmin((double)n_cells_max, fmax(1.0, (double)n / kTargetPerCell));
  double h = emax > 0.f ? (double)emax : 1.0;
  if (emax > 0.f && cells_at(ext, d, h * 1e-9) > want) {  // bisection on the cell edge: cells_at is non-increasing in h
    double lo_h = h * 1e-9, hi_h = h;                      // cells_at(hi_h) == 1 <= want
    for (int it = 0; it < 64; ++it) {
      const double mid = 0.5 * (lo_h + hi_h);
      if (cells_at(ext, d, mid) > want) lo_h = mid;
      else hi_h = mid;
    }
    h = hi_h;
  }
  int64_t total = 1;
  for (int j = 0; j < 3; ++j) {
    int dj = (j < d && ext[j] > 0.f) ? (int)fmax(1.0, ceil((double)ext[j] / h)) : 1;
    P->dim[j] = dj;
    total *= dj;
  }
  if (total > n_cells_max) {  // (cannot happen: the bisection keeps cells_at(h) <= want <= n_cells_max; a single cell is always correct)
    P->dim[0] = P->dim[1] = P->dim[2] = 1;
    total = 1;
  }
  P->h = (float)h;
  P->inv_h = (float)(1.0 / h);
  P->n_cells = (int)total;
  // cell assignment and face positions are f32 expressions of coordinates up to maxabs + emax: a few ulps of that, generously
  P->slack = 32.f * 1.1920929e-7f * (maxabs + emax + (float)h);
}

__device__ __forceinline__ void cell_of(const GridParams& P, float x, float y, float z, int& cx, int& cy, int& cz) {
  cx = min(max((int)floorf((x - P.lo[0]) * P.inv_h), 0), P.dim[0] - 1);
  cy = min(max((int)floorf((y - P.lo[1]) * P.inv_h), 0), P.dim[1] - 1);
  cz = min(max((int)floorf((z - P.lo[2]) * P.inv_h), 0), P.dim[2] - 1);
}

__global__ __launch_bounds__(256) void grid_count_kernel(int64_t n, int d, const float* __restrict__ X, int64_t ldx, const GridParams* __restrict__ Pp,
                                                         int32_t* __restrict__ cell, int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const GridParams P = *Pp;
  const float x = X[i * ldx], y = d > 1 ? X[i * ldx + 1] : 0.f, z = d > 2 ? X[i * ldx + 2] : 0.f;
  int cx, cy, cz;
  cell_of(P, x, y, z, cx, cy, cz);
  const int id = (cz * P.dim[1] + cy) * P.dim[0] + cx;
  cell[i] = id;
  atomicAdd(count + id, 1);
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(int64_t n, int d, const float* __restrict__ X, int64_t ldx, const int32_t* __restrict__ cell,
                                                           const int32_t* __restrict__ start, int32_t* __restrict__ cursor, float4* __restrict__ sorted) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int id = cell[i];
  const int pos = start[id] + atomicAdd(cursor + id, 1);
  sorted[pos] = make_float4(X[i * ldx], d > 1 ? X[i * ldx + 1] : 0.f, d > 2 ? X[i * ldx + 2] : 0.f, __int_as_float((int)i));
}

__device__ __forceinline__ bool before(float d2, int idx, float od, int oi) { return d2 < od || (d2 == od && idx < oi); }

// per-lane unsorted list in LDS ([k][QB], element s of lane t at [s * QB + t]) with its worst entry tracked, as in knn.hip
struct List {
  float* d;
  int* i;
};
__device__ __forceinline__ void rescan(const List& L, int k, float& tau_d, int& tau_i, int& tau_pos) {
  float md = L.d[0];
  int mi = L.i[0], mp = 0;
  for (int s = 1; s < k; ++s) {
    const float dd = L.d[s * QB];
    const int ii = L.i[s * QB];
    if (before(md, mi, dd, ii)) {
      md = dd;
      mi = ii;
      mp = s;
    }
  }
  tau_d = md;
  tau_i = mi;
  tau_pos = mp;
}

__device__ __forceinline__ float d2_chain(int d, float qx, float qy, float qz, const float4& p) {
  // the defined chain: feature order, every operation rounded separately (knn.hip's contract)
  float diff = __fsub_rn(qx, p.x);
  float acc = __fadd_rn(0.f, __fmul_rn(diff, diff));
  if (d > 1) {
    diff = __fsub_rn(qy, p.y);
    acc = __fadd_rn(acc, __fmul_rn(diff, diff));
  }
  if (d > 2) {
    diff = __fsub_rn(qz, p.z);
    acc = __fadd_rn(acc, __fmul_rn(diff, diff));
  }
  return acc;
}

__global__ __launch_bounds__(QB) void grid_query_kernel(int64_t n, int d, int64_t q_begin, int64_t nq, int k, const GridParams* __restrict__ Pp,
                                                        const int32_t* __restrict__ start, const float4* __restrict__ sorted,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ld = reinterpret_cast<float*>(smem);
  int* li = reinterpret_cast<int*>(ld + k * QB);
  const int tid = threadIdx.x;
  const int64_t t = (int64_t)blockIdx.x * QB + tid;
  if (t >= n) return;
  const float4 me = sorted[t];
  const int64_t q = (int64_t)__float_as_int(me.w);
  if (q < q_begin || q >= q_begin + nq) return;
  const GridParams P = *Pp;
  List L{ld + tid, li + tid};
  int cnt = 0, tau_i = 0x7fffffff, tau_pos = 0;
  float tau_d = INFINITY;
  int cx, cy, cz;
  cell_of(P, me.x, me.y, me.z, cx, cy, cz);
  const int rmax = max(P.dim[0], max(P.dim[1], P.dim[2]));
  bool done = false;
  for (int r = 0; r <= rmax && !done; ++r) {
    if (r > kMaxRing) {  // an outlier whose neighbours are many cells away: everything, once (ring order no longer matters)
      cnt = 0;
      tau_d = INFINITY;
      tau_i = 0x7fffffff;
      for (int64_t p = 0; p < n; ++p) {
        const float4 c = sorted[p];
        const float d2 = d2_chain(d, me.x, me.y, me.z, c);
        const int idx = __float_as_int(c.w);
        if (cnt < k || before(d2, idx, tau_d, tau_i)) {
          const int pos = cnt < k ? cnt : tau_pos;
          L.d[pos * QB] = d2;
          L.i[pos * QB] = idx;
          if (cnt < k) ++cnt;
          if (cnt == k) rescan(L, k, tau_d, tau_i, tau_pos);
        }
      }
      break;
    }
    const int z0 = max(cz - r, 0), z1 = min(cz + r, P.dim[2] - 1), y0 = max(cy - r, 0), y1 = min(cy + r, P.dim[1] - 1);
    const int x0 = max(cx - r, 0), x1 = min(cx + r, P.dim[0] - 1);
    for (int zc = z0; zc <= z1; ++zc)
      for (int yc = y0; yc <= y1; ++yc) {
        const bool shell_zy = abs(zc - cz) == r || abs(yc - cy) == r;  // this (z, y) row lies on the ring's surface: all its x cells belong to ring r
        const int row = (zc * P.dim[1] + yc) * P.dim[0];
        if (shell_zy) {
          // consecutive cells of a row are consecutive in the sorted array: one contiguous run
          const int s = start[row + x0], e = start[row + x1 + 1];
          for (int p = s; p < e; ++p) {
            const float4 c = sorted[p];
            const float d2 = d2_chain(d, me.x, me.y, me.z, c);
            const int idx = __float_as_int(c.w);
            if (cnt < k || before(d2, idx, tau_d, tau_i)) {
              const int pos = cnt < k ? cnt : tau_pos;
              L.d[pos * QB] = d2;
              L.i[pos * QB] = idx;
              if (cnt < k) ++cnt;
              if (cnt == k) rescan(L, k, tau_d, tau_i, tau_pos);
            }
          }
        } else {  // interior row of the block: only its two end cells are new
#pragma unroll
          for (int side = 0; side < 2; ++side) {
            const int xc = side ? cx + r : cx - r;
            if (xc < 0 || xc >= P.dim[0] || (side && r == 0)) continue;
            const int s = start[row + xc], e = start[row + xc + 1];
            for (int p = s; p < e; ++p) {
              const float4 c = sorted[p];
              const float d2 = d2_chain(d, me.x, me.y, me.z, c);
              const int idx = __float_as_int(c.w);
              if (cnt < k || before(d2, idx, tau_d, tau_i)) {
                const int pos = cnt < k ? cnt : tau_pos;
                L.d[pos * QB] = d2;
                L.i[pos * QB] = idx;
                if (cnt < k) ++cnt;
                if (cnt == k) rescan(L, k, tau_d, tau_i, tau_pos);
              }
            }
          }
        }
      }
    // every unexamined point lies beyond a face of the examined block that is not the grid's own boundary
    float bound = INFINITY;
    const float pos3[3] = {me.x, me.y, me.z};
    const int c3[3] = {cx, cy, cz};
    for (int j = 0; j < 3; ++j) {
      if (c3[j] - r > 0) bound = fminf(bound, pos3[j] - (P.lo[j] + (float)(c3[j] - r) * P.h));
      if (c3[j] + r < P.dim[j] - 1) bound = fminf(bound, (P.lo[j] + (float)(c3[j] + r + 1) * P.h) - pos3[j]);
    }
    if (bound == INFINITY) done = true;  // the block is the whole grid
    else if (cnt == k) {
      const float b = bound - P.slack;
      if (b > 0.f && tau_d < b * b * 0.999999f) done = true;
    }
  }
  // selection sort by (d2, index) and write
  int32_t* oi = out_idx + (q - q_begin) * k;
  float* od = out_dist + (q - q_begin) * k;
  for (int s = 0; s < k; ++s) {
    if (s < cnt) {
      float bd = L.d[s * QB];
      int bi = L.i[s * QB], bp = s;
      for (int r2 = s + 1; r2 < cnt; ++r2) {
        const float rd = L.d[r2 * QB];
        const int ri = L.i[r2 * QB];
        if (before(rd, ri, bd, bi)) {
          bd = rd;
          bi = ri;
          bp = r2;
        }
      }
      if (bp != s) {
        L.d[bp * QB] = L.d[s * QB];
        L.i[bp * QB] = L.i[s * QB];
      }
      oi[s] = bi;
      od[s] = (float)sqrt((double)bd);
    } else {
      oi[s] = -1;
      od[s] = INFINITY;
    }
  }
}

size_t r256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

namespace dh {

bool knn_grid_supported(int64_t d, int k) { return d >= 1 && d <= 3 && k >= 1 && k <= 32; }
bool knn_grid_applies(int64_t n, int64_t d, int k) { return knn_grid_supported(d, k) && n >= 2048; }  // (below that the scan's one launch wins)

size_t knn_grid_workspace_bytes(int64_t n) {
  // params, bbox, cell[n], count[n + 1], start[n + 2], cursor[n], sorted[n] float4, scan workspace
  return r256(sizeof(GridParams)) + r256(32) + r256((size_t)n * 4) + r256((size_t)(n + 1) * 4) + r256((size_t)(n + 2) * 4) + r256((size_t)(n + 1) * 4) +
         r256((size_t)n * 16) + r256(dh_exclusive_scan_i32_workspace_bytes(n + 1));
}

int knn_grid_launch(int64_t n, int d, const float* X, int64_t ldx, int64_t q_begin, int64_t nq, int k, int32_t* out_idx, float* out_dist, void* workspace,
                    hipStream_t st) {
  char* ws = static_cast<char*>(workspace);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = ws + off;
    off += r256(bytes);
    return p;
  };
  GridParams* P = reinterpret_cast<GridParams*>(take(sizeof(GridParams)));
  unsigned* bbox = reinterpret_cast<unsigned*>(take(32));
  int32_t* cell = reinterpret_cast<int32_t*>(take((size_t)n * 4));
  int32_t* count = reinterpret_cast<int32_t*>(take((size_t)(n + 1) * 4));
  int32_t* start = reinterpret_cast<int32_t*>(take((size_t)(n + 2) * 4));
  int32_t* cursor = reinterpret_cast<int32_t*>(take((size_t)(n + 1) * 4));
  float4* sorted = reinterpret_cast<float4*>(take((size_t)n * 16));
  const size_t scan_bytes = dh_exclusive_scan_i32_workspace_bytes(n + 1);
  void* scan_ws = take(scan_bytes);
  const int64_t n_cells_max = n;
  const unsigned g1 = (unsigned)std::min<int64_t>(ceil_div(n + 1, 256), 2048);
  hipLaunchKernelGGL(grid_init_kernel, dim3(g1), dim3(256), 0, st, n_cells_max + 1, bbox, count, cursor);
  hipLaunchKernelGGL(grid_bbox_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 1024)), dim3(256), 0, st, n, d, X, ldx, bbox);
  hipLaunchKernelGGL(grid_params_kernel, dim3(1), dim3(1), 0, st, n, d, n_cells_max, bbox, P);
  hipLaunchKernelGGL(grid_count_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, d, X, ldx, P, cell, count);
  const int rc = dh_exclusive_scan_i32(n_cells_max + 1, count, start, scan_ws, scan_bytes, reinterpret_cast<dh_stream_t>(st));  // start[0 .. n_cells_max + 1]
  if (rc != DH_OK) return rc;
  hipLaunchKernelGGL(grid_scatter_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n, d, X, ldx, cell, start, cursor, sorted);
  hipLaunchKernelGGL(grid_query_kernel, dim3((unsigned)ceil_div(n, QB)), dim3(QB), (size_t)k * QB * 8, st, n, d, q_begin, nq, k, P, start, sorted, out_idx,
                     out_dist);
  return check_launch("dh_knn_bruteforce_f32 (grid)");
}

}  // namespace dh
