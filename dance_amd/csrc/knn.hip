// Exact brute-force kNN (rows A11/A12/A13 of §8a): the public entry point, the vector-ALU SCAN and
// the sample scan of the matrix-core filter path (knn_filter.hip).
//
// Parity contract ("bit-exact graph indices"): the squared distance of a pair is DEFINED as
//     d2 = (((0 + sq(x_0-y_0)) + sq(x_1-y_1)) + ...),  sq(u) = rn(u*u), every op a separate f32
// round-to-nearest operation in feature order — exactly what the numpy oracle evaluates — and the
// neighbours of a query are the k smallest (d2, index) pairs in lexicographic order (ties go to
// the lower index; the query itself is its own nearest neighbour with d2 = 0, as in sklearn and
// scanpy).  A |x|^2 - 2xy expansion does not reproduce those bits, so the scan evaluates the chain itself on
// the VALUs; the filter path uses the expansion on the matrix cores only to discard pairs and re-ranks the
// survivors with this same chain.
//
// Scan mapping: one lane per query (256 queries per block), the query's features in registers.  d <= 64: candidate
// rows are wave-uniform, so they arrive through scalar loads as SGPR operands (knn_sreg_kernel).  d > 64: features
// in blocks of 32 per 32 candidates, the candidates' features again through scalar loads (knn_big_kernel).  Each lane keeps its
// current k-th best (tau) in registers; a candidate that beats tau replaces the worst entry of the lane's unsorted
// list (LDS for k <= 32, else the output arrays themselves).  For random data a query sees only ~k ln(N/k)
// insertions, so the kernels are bound by the 3 VALU ops per (pair, feature).
#include "common.h"
#include "knn_fold.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int QB = 256;  // queries per block
constexpr int KLDS = 32; // largest k whose lists live in LDS

struct List {
  float* d;  // element s of this lane's list at d[s * stride]
  int* i;
  int64_t stride;
};

__device__ __forceinline__ bool before(float d2, int idx, float od, int oi) {
  return d2 < od || (d2 == od && idx < oi);
}

// The per-lane list is kept UNSORTED with its worst entry (tau = the current k-th best) tracked by position:
// an accepted candidate overwrites the worst entry and the list is rescanned for the new worst — k independent
// LDS reads that pipeline, instead of a sorted insertion's chain of dependent read-compare-write steps.  While one
// lane of a wavefront inserts, the other 63 wait, so this latency is what the scan pays per accepted candidate.
__device__ __forceinline__ void rescan(const List& L, int k, float& tau_d, int& tau_i, int& tau_pos) {
  float md = L.d[0];
  int mi = L.i[0], mp = 0;
  for (int s = 1; s < k; s += 4) {
    float dd[4];
    int ii[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ss = min(s + u, k - 1);
      dd[u] = L.d[ss * L.stride];
      ii[u] = L.i[ss * L.stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (before(md, mi, dd[u], ii[u])) { md = dd[u]; mi = ii[u]; mp = min(s + u, k - 1); }
  }
  tau_d = md; tau_i = mi; tau_pos = mp;
}

__device__ __forceinline__ void insert(const List& L, int k, int& cnt, float& tau_d, int& tau_i, int& tau_pos, float d2, int idx) {
  const int pos = cnt < k ? cnt : tau_pos;
  L.d[pos * L.stride] = d2;
  L.i[pos * L.stride] = idx;
  if (cnt < k) ++cnt;
  if (cnt == k) rescan(L, k, tau_d, tau_i, tau_pos);
}

// Selection-sorts the lane's list in place by (d2, index) and writes indices + distances (or raw d2 for a partial
// list that knn_merge_kernel will finish).
__device__ __forceinline__ void finish(const List& L, bool lds_list, int k, int cnt, int64_t q_local, bool valid,
                                       int32_t* __restrict__ out_idx, float* __restrict__ out_dist, bool raw_d2 = false) {
  if (!valid) return;
  int32_t* oi = out_idx + q_local * k;
  float* od = out_dist + q_local * k;
  for (int s = 0; s < k; ++s) {
    if (s < cnt) {
      float bd = L.d[s * L.stride];
      int bi = L.i[s * L.stride], bp = s;
      for (int r = s + 1; r < cnt; ++r) {
        const float rd = L.d[r * L.stride];
        const int ri = L.i[r * L.stride];
        if (before(rd, ri, bd, bi)) { bd = rd; bi = ri; bp = r; }
      }
      if (bp != s) {  // move the displaced entry into the hole
        L.d[bp * L.stride] = L.d[s * L.stride];
        L.i[bp * L.stride] = L.i[s * L.stride];
      }
      oi[s] = bi;
      od[s] = raw_d2 ? bd : (float)sqrt((double)bd);  // f64 sqrt then one rounding == correctly rounded f32 sqrt
    } else {  // fewer than k points exist
      oi[s] = -1;
      od[s] = __int_as_float(0x7f800000);
    }
  }
  (void)lds_list;
}

// d <= 64, scalar-operand form: a candidate row is the same for every lane of a wavefront, so it belongs in
// SGPRs, not in LDS — the rows of the zero-padded copy Xp [n][DCH] are fetched with scalar loads (s_load_dwordx8/16
// through the scalar cache, shared by the block's 4 waves) and enter the VALU as the one SGPR operand of
// v_sub_f32.  No LDS staging, no barriers in the scan; the per-lane lists stay in LDS.
// (Scoring two candidates per lane with v_pk_add_f32 / v_pk_mul_f32 on a pair-interleaved copy was measured at the
// same rate — packed f32 issues at half the rate of plain f32 on gfx950 — so the plain form is kept.)
template <int DCH>
__global__ __launch_bounds__(QB) void knn_sreg_kernel(int64_t n, const float* __restrict__ Xp, const float* __restrict__ Cp,
                                                      int64_t q_begin, int64_t q_end, int k, bool lds_list, bool raw_d2,
                                                      int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  const int64_t cand_lo = n * blockIdx.y / gridDim.y, cand_hi = n * (blockIdx.y + 1) / gridDim.y;
  const int64_t nq_all = q_end - q_begin;
  if (gridDim.y > 1) { out_idx += (int64_t)blockIdx.y * nq_all * k; out_dist += (int64_t)blockIdx.y * nq_all * k; }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ld = reinterpret_cast<float*>(smem);                      // [k][QB] when lds_list
  int* li = reinterpret_cast<int*>(ld + (lds_list ? k * QB : 0));  // [k][QB]

  const int tid = threadIdx.x;
  const int64_t q_local = (int64_t)blockIdx.x * QB + tid;
  const int64_t q = q_begin + q_local;
  const bool valid = q < q_end;

  constexpr int RS = DCH;  // row stride of Xp (padding rows to whole 64-byte lines measured slower: the scan is bound by scalar-cache bytes)
  float x[DCH];
  {
    const f32x4* xq = reinterpret_cast<const f32x4*>(Xp + (valid ? q : 0) * RS);
#pragma unroll
    for (int t = 0; t < DCH / 4; ++t) {
      const f32x4 v = xq[t];
      x[t * 4] = v[0]; x[t * 4 + 1] = v[1]; x[t * 4 + 2] = v[2]; x[t * 4 + 3] = v[3];
    }
  }
  List L;
  if (lds_list) { L.d = ld + tid; L.i = li + tid; L.stride = QB; }
  else { L.d = out_dist + (valid ? q_local : 0) * k; L.i = out_idx + (valid ? q_local : 0) * k; L.stride = 1; }
  int cnt = 0, tau_i = 0x7fffffff, tau_pos = 0;
  float tau_d = __int_as_float(0x7f800000);

  for (int64_t c = cand_lo; c < cand_hi; ++c) {
    const float* __restrict__ y = Cp + c * RS;  // wave-uniform address -> scalar loads
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < DCH; ++t) {
      const float diff = __fsub_rn(x[t], y[t]);
      acc = __fadd_rn(acc, __fmul_rn(diff, diff));
    }
    const int idx = (int)c;
    if (valid && (cnt < k || before(acc, idx, tau_d, tau_i))) insert(L, k, cnt, tau_d, tau_i, tau_pos, acc, idx);
  }
  finish(L, lds_list, k, cnt, q_local, valid, out_idx, out_dist, raw_d2 || gridDim.y > 1);
}

// Xp[r][0:RS] = X[r][0:d] followed by zeros (zero features add exact zeros to every distance)
__global__ __launch_bounds__(256) void knn_pad_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx, int rs,
                                                      float* __restrict__ Xp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * rs) return;
  const int64_t r = i / rs;
  const int t = (int)(i % rs);
  Xp[i] = t < d ? X[r * ldx + t] : 0.f;
}

// d > 64: the query row no longer fits in registers, so the scan is blocked: CT candidates x FC features at a time.
// Per block of work a lane loads FC features of its query (one 128-byte line), then for each of the CT candidates the
// matching FC features arrive through scalar loads (wave-uniform row, SGPR operands as in knn_sreg_kernel) and extend
// that candidate's accumulator — features strictly in ascending order for every pair, so the chain is the defined one.
// No LDS staging and no barriers (the previous LDS-tiled form ran at 0.21 of the VALU rate).
__global__ __launch_bounds__(QB) void knn_big_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                     const float* __restrict__ C, int64_t ldc, int64_t q_begin, int64_t q_end,
                                                     int k, bool lds_list, bool raw_d2, int32_t* __restrict__ out_idx,
                                                     float* __restrict__ out_dist) {
  constexpr int CT = 32, FC = 32;
  const int64_t cand_lo = n * blockIdx.y / gridDim.y, cand_hi = n * (blockIdx.y + 1) / gridDim.y;
  const int64_t nq_all = q_end - q_begin;
  if (gridDim.y > 1) { out_idx += (int64_t)blockIdx.y * nq_all * k; out_dist += (int64_t)blockIdx.y * nq_all * k; }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ld = reinterpret_cast<float*>(smem);
  int* li = reinterpret_cast<int*>(ld + (lds_list ? k * QB : 0));

  const int tid = threadIdx.x;
  const int64_t q_local = (int64_t)blockIdx.x * QB + tid;
  const int64_t q = q_begin + q_local;
  const bool valid = q < q_end;
  const float* xq = X + (valid ? q : 0) * ldx;

  List L;
  if (lds_list) { L.d = ld + tid; L.i = li + tid; L.stride = QB; }
  else { L.d = out_dist + (valid ? q_local : 0) * k; L.i = out_idx + (valid ? q_local : 0) * k; L.stride = 1; }
  int cnt = 0, tau_i = 0x7fffffff, tau_pos = 0;
  float tau_d = __int_as_float(0x7f800000);

  const int64_t d_full = d / FC * FC;
  for (int64_t c0 = cand_lo; c0 < cand_hi; c0 += CT) {
    const int lim = (int)min((int64_t)CT, cand_hi - c0);  // wave-uniform
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    for (int64_t t0 = 0; t0 < d_full; t0 += FC) {
      float x[FC];
#pragma unroll
      for (int t = 0; t < FC; ++t) x[t] = xq[t0 + t];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c < lim) {
          const float* __restrict__ y = C + (c0 + c) * ldc + t0;  // wave-uniform address -> scalar loads
#pragma unroll
          for (int t = 0; t < FC; ++t) {
            const float diff = __fsub_rn(x[t], y[t]);
            acc[c] = __fadd_rn(acc[c], __fmul_rn(diff, diff));
          }
        }
      }
    }
    if (d_full < d) {  // last, partial feature block: guarded element by element (uniform guards)
      float x[FC];
#pragma unroll
      for (int t = 0; t < FC; ++t) x[t] = (d_full + t < d) ? xq[d_full + t] : 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c < lim) {
          const float* __restrict__ y = C + (c0 + c) * ldc + d_full;
#pragma unroll
          for (int t = 0; t < FC; ++t) {
            if (d_full + t < d) {
              const float diff = __fsub_rn(x[t], y[t]);
              acc[c] = __fadd_rn(acc[c], __fmul_rn(diff, diff));
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int64_t ci = c0 + c;
      if (valid && c < lim && (cnt < k || before(acc[c], (int)ci, tau_d, tau_i)))
        insert(L, k, cnt, tau_d, tau_i, tau_pos, acc[c], (int)ci);
    }
  }
  finish(L, lds_list, k, cnt, q_local, valid, out_idx, out_dist, raw_d2 || gridDim.y > 1);
}

// Merge P partial (d2, idx) lists per query (each sorted ascending) into the final k smallest by (d2, idx): one
// thread per query, P cursors, k selection steps — deterministic, the same order a single scan produces.
__global__ __launch_bounds__(256) void knn_merge_kernel(int64_t nq, int k, int P, bool raw_d2, const int32_t* __restrict__ part_idx,
                                                        const float* __restrict__ part_d2, int32_t* __restrict__ out_idx,
                                                        float* __restrict__ out_dist) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  constexpr int MAXP = 64;
  unsigned char cur[MAXP];
  for (int p = 0; p < P; ++p) cur[p] = 0;
  for (int s = 0; s < k; ++s) {
    int best = -1, bi = -1;
    float bd = 0.f;
    for (int p = 0; p < P; ++p) {
      if (cur[p] >= k) continue;
      const int64_t off = ((int64_t)p * nq + q) * k + cur[p];
      const int ci = part_idx[off];
      if (ci < 0) continue;  // this slice ran out of points
      const float cd = part_d2[off];
      if (best < 0 || before(cd, ci, bd, bi)) { best = p; bd = cd; bi = ci; }
    }
    if (best < 0) {
      out_idx[q * k + s] = -1;
      out_dist[q * k + s] = __int_as_float(0x7f800000);
    } else {
      out_idx[q * k + s] = bi;
      out_dist[q * k + s] = raw_d2 ? bd : (float)sqrt((double)bd);
      ++cur[best];
    }
  }
}

}  // namespace

namespace {

// candidate slices per query block so that >= ~2048 blocks (8 per CU) are in flight; k <= 255 for the merge cursors
int knn_slices(int64_t nq, int k) {
  const int64_t blocks = dh::ceil_div(nq, QB);
  if (blocks >= 1024 || k > 255) return 1;
  int64_t p = dh::ceil_div(2048, blocks);
  return (int)(p > 64 ? 64 : p);
}

int knn_padded_width(int64_t d) { return d <= 64 ? (int)((d + 3) / 4 * 4) : 0; }
size_t round64(size_t b) { return (b + 63) / 64 * 64; }

// Workspace carve-up shared by the size query and the launcher (offsets in bytes, every region 64-byte aligned).
struct Layout {
  int algo;                      // resolved: DH_KNN_SCAN or DH_KNN_FILTER
  int P;                         // candidate slices of the scan (of the sample scan in filter mode)
  int dch;                       // padded width of the register kernel (0: d > 64)
  bool fold;                     // filter: the fp16 two-pass path (d <= 64)
  dh::KnnFoldGeom g;             // its geometry
  int64_t S, stride;             // sample rows j * stride, j < S (filter)
  int cap;                       // survivor list capacity per query (filter)
  size_t partial, xp, xs, mean, scale, norms, rq, cn, a2, b2, counts, surv, total;
};

Layout make_layout(int64_t n, int64_t d, int64_t nq, int k, int algo) {
  Layout L{};
  // auto: the filter pays off once the n x n pair count dwarfs its fixed passes; it needs k <= 64 (one carried key per lane)
  // (spatial coordinates, d <= 3: a cell grid examines ~100 pairs per query instead of n — knn_grid.hip)
  if (algo == DH_KNN_AUTO) algo = dh::knn_grid_applies(n, d, k) ? DH_KNN_GRID : (n >= 16384 && nq >= 1024 && k <= 64) ? DH_KNN_FILTER : DH_KNN_SCAN;
  L.algo = algo;
  if (algo == DH_KNN_GRID) {
    L.total = dh::knn_grid_workspace_bytes(n);
    return L;
  }
  L.dch = knn_padded_width(d);
  L.P = knn_slices(nq, k);
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += round64(bytes); return o; };
  L.partial = take(L.P > 1 ? (size_t)L.P * nq * k * 8 : 0);
  L.xp = take(L.dch ? (size_t)n * L.dch * sizeof(float) : 0);
  if (algo == DH_KNN_FILTER) {
    L.fold = dh::knn_fold_applies(d);
    L.mean = take(dh::knn_filter_mean_floats(d) * sizeof(float));
    L.norms = take((size_t)n * sizeof(float));
    if (L.fold) {
      L.g = dh::knn_fold_geom(n, d, nq, k);
      L.S = L.g.S;
      L.stride = L.g.stride0;
      L.cap = L.g.cap;
      L.scale = take(64);
      L.a2 = take((size_t)n * L.g.K3 * 2);
      L.b2 = take((size_t)L.g.n_pos * L.g.K3 * 2);
      L.counts = take((size_t)nq * L.g.n_seg_total * sizeof(int32_t));
    } else {
      L.S = dh::knn_filter_sample_size(n);
      L.stride = n / L.S;
      int n_seg, seg;
      dh::knn_filter_geometry(n, d, nq, k, &n_seg, &seg);
      L.cap = n_seg * seg;
      const size_t K3 = (size_t)dh::knn_filter_k3(d);
      L.rq = take((size_t)nq * sizeof(float));
      L.cn = take((size_t)n * sizeof(float));
      L.a2 = take((size_t)n * K3 * 2);
      L.b2 = take((size_t)n * K3 * 2);
      L.counts = take((size_t)nq * n_seg * sizeof(int32_t));
    }
    L.xs = take((size_t)L.S * (L.dch ? L.dch : d) * sizeof(float));
    L.surv = take((size_t)nq * L.cap * sizeof(int32_t));
  }
  L.total = off;
  return L;
}

// Exact scan of queries [q_begin, q_end) (rows of Q) against the n_cand rows of C.  For d <= 64, Q and C are the
// zero-padded copies (row stride dch); otherwise plain matrices with their leading dimensions.
void scan_launch(int64_t n_cand, int64_t d, int dch, const float* Q, int64_t ldq, const float* C, int64_t ldc, int64_t q_begin,
                 int64_t q_end, int k, int P, bool raw_d2, void* partial, int32_t* out_idx, float* out_dist, hipStream_t st) {
  const int64_t nq = q_end - q_begin;
  int32_t* k_idx = out_idx;
  float* k_dist = out_dist;
  if (P > 1) {
    k_idx = static_cast<int32_t*>(partial);
    k_dist = reinterpret_cast<float*>(k_idx + (size_t)P * nq * k);
  }
  // per-lane lists live in LDS for small k, otherwise in the output (or partial) arrays themselves
  const bool lds_list = k <= KLDS;
  const size_t list_bytes = lds_list ? (size_t)k * QB * 8 : 0;
  dim3 grid((unsigned)dh::ceil_div(nq, QB), (unsigned)P), block(QB);
#define DH_KNN_SMALL(DCH)                                                                                             \
  case DCH:                                                                                                           \
    hipLaunchKernelGGL(knn_sreg_kernel<DCH>, grid, block, list_bytes, st, n_cand, Q, C, q_begin, q_end, k, lds_list,  \
                       raw_d2, k_idx, k_dist);                                                                        \
    break
  // the query row is held in registers, padded to a multiple of 4 features (zero padding adds exact zeros to the
  // distance); one instantiation per padded width so that d = 50 does 52, not 64, features of work per pair
  switch (dch) {
    DH_KNN_SMALL(4); DH_KNN_SMALL(8); DH_KNN_SMALL(12); DH_KNN_SMALL(16); DH_KNN_SMALL(20); DH_KNN_SMALL(24);
    DH_KNN_SMALL(28); DH_KNN_SMALL(32); DH_KNN_SMALL(36); DH_KNN_SMALL(40); DH_KNN_SMALL(44); DH_KNN_SMALL(48);
    DH_KNN_SMALL(52); DH_KNN_SMALL(56); DH_KNN_SMALL(60); DH_KNN_SMALL(64);
    default:
      hipLaunchKernelGGL(knn_big_kernel, grid, block, list_bytes, st, n_cand, d, Q, ldq, C, ldc,
                         q_begin, q_end, k, lds_list, raw_d2, k_idx, k_dist);
  }
#undef DH_KNN_SMALL
  if (P > 1)
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)dh::ceil_div(nq, 256)), dim3(256), 0, st, nq, k, P, raw_d2, k_idx, k_dist,
                       out_idx, out_dist);
}

}  // namespace

extern "C" size_t dh_knn_bruteforce_f32_workspace_bytes(int64_t n, int64_t d, int64_t n_queries, int k, int algo) {
  if (n <= 0 || n_queries <= 0 || k <= 0) return 0;
  return make_layout(n, d, n_queries, k, algo).total;
}

extern "C" int dh_knn_filter_plan(int64_t n, int64_t d, int64_t n_queries, int k, int64_t* out, int n_out) {
  constexpr int kFields = 26;
  if (!out || n_out <= 0 || n <= 0 || n_queries <= 0 || k <= 0) return kFields;
  int64_t f[kFields] = {0};
  if (dh::knn_fold_applies(d)) {
    const dh::KnnFoldGeom g = dh::knn_fold_geom(n, d, n_queries, k);
    f[0] = g.n_pass; f[1] = g.G; f[2] = g.H; f[3] = g.qmagic; f[4] = g.n1; f[5] = g.n_pos; f[6] = g.S; f[7] = g.stride0;
    f[8] = g.K3; f[9] = g.cap; f[10] = g.tiles;
    for (int p = 0; p < 3; ++p) {
      f[11 + 5 * p] = g.t_begin[p]; f[12 + 5 * p] = g.t_end[p]; f[13 + 5 * p] = g.tps[p];
      f[14 + 5 * p] = g.n_seg[p]; f[15 + 5 * p] = g.seg[p];
    }
  }
  for (int i = 0; i < kFields && i < n_out; ++i) out[i] = f[i];
  return kFields;
}

extern "C" int dh_knn_bruteforce_f32(int64_t n, int64_t d, const float* X, int64_t ldx, int64_t q_begin,
                                     int64_t q_end, int k, int algo, int32_t* out_idx, float* out_dist, void* workspace,
                                     size_t workspace_bytes, dh_stream_t stream) {
  if (n < 0 || d < 0 || k < 0) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: negative size");
  if (q_begin < 0 || q_end > n || q_begin > q_end)
    return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: bad query range [%lld, %lld)", (long long)q_begin, (long long)q_end);
  if (algo != DH_KNN_AUTO && algo != DH_KNN_SCAN && algo != DH_KNN_FILTER && algo != DH_KNN_GRID)
    return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: bad algo %d", algo);
  if (algo == DH_KNN_GRID && !dh::knn_grid_supported(d, k)) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: the grid path needs d <= 3 and k <= 32");
  if (q_end == q_begin || k == 0) return DH_OK;
  if (!X || !out_idx || !out_dist) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: null pointer");
  if (ldx < d) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: ldx < d");
  if (n >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: n >= 2^31");
  if (algo == DH_KNN_FILTER && k > 64) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: the filter path needs k <= 64");
  hipStream_t st = dh::as_stream(stream);
  const int64_t nq = q_end - q_begin;
  const Layout L = make_layout(n, d, nq, k, algo);
  if (L.total && (!workspace || workspace_bytes < L.total || (reinterpret_cast<uintptr_t>(workspace) & 63u)))
    return dh::fail(DH_ERR_WORKSPACE, "dh_knn_bruteforce_f32: workspace %zu < %zu bytes (or not 64-byte aligned)", workspace_bytes, L.total);
  if (L.algo == DH_KNN_GRID) return dh::knn_grid_launch(n, (int)d, X, ldx, q_begin, nq, k, out_idx, out_dist, workspace, st);
  char* ws = static_cast<char*>(workspace);
  float* Xp = reinterpret_cast<float*>(ws + L.xp);
  if (L.dch)
    hipLaunchKernelGGL(knn_pad_kernel, dim3((unsigned)dh::ceil_div(n * L.dch, 256)), dim3(256), 0, st, n, d, X, ldx, L.dch, Xp);
  const float* Q = L.dch ? Xp : X;          // what the scan kernels read
  const int64_t ldq = L.dch ? L.dch : ldx;
  if (L.algo == DH_KNN_SCAN) {
    scan_launch(n, d, L.dch, Q, ldq, Q, ldq, q_begin, q_end, k, L.P, false, ws + L.partial, out_idx, out_dist, st);
    return dh::check_launch("dh_knn_bruteforce_f32");
  }
  // filter: (1) exact k-th distance inside a strided sample -> out_dist[:, k-1] (raw d2), (2) + (3) in knn_filter.hip
  float* Xs = reinterpret_cast<float*>(ws + L.xs);
  const int rs = L.dch ? L.dch : (int)d;
  dh::knn_filter_sample_strided(L.S, L.stride, d, X, ldx, rs, Xs, st);
  scan_launch(L.S, d, L.dch, Q, ldq, Xs, rs, q_begin, q_end, k, L.P, true, ws + L.partial, out_idx, out_dist, st);
  if (L.fold)
    return dh::knn_fold_launch(L.g, n, d, X, ldx, Q, ldq, L.dch, q_begin, nq, k, reinterpret_cast<float*>(ws + L.mean),
                               reinterpret_cast<unsigned int*>(ws + L.scale), ws + L.a2, ws + L.b2, reinterpret_cast<float*>(ws + L.norms),
                               reinterpret_cast<int32_t*>(ws + L.counts), reinterpret_cast<int32_t*>(ws + L.surv), out_idx, out_dist, st);
  return dh::knn_filter_launch(n, d, X, ldx, Q, ldq, L.dch ? L.dch : d, q_begin, nq, k, out_dist, reinterpret_cast<float*>(ws + L.mean),
                               reinterpret_cast<uint16_t*>(ws + L.a2), reinterpret_cast<uint16_t*>(ws + L.b2),
                               reinterpret_cast<float*>(ws + L.norms), reinterpret_cast<float*>(ws + L.rq),
                               reinterpret_cast<float*>(ws + L.cn), reinterpret_cast<int32_t*>(ws + L.counts),
                               reinterpret_cast<int32_t*>(ws + L.surv), out_idx, out_dist, st);
}
