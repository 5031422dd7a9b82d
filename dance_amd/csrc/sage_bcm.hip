// AdaptiveSAGE cell <- gene aggregation on the matrix cores, TWO wavefronts per SIMD (round 4; the unsplit path of
// dh_sage_window_mfma — see sage_mfma.hip for the arithmetic and the precision argument, which are unchanged:
// dance/models/nn/gnn.py:62-90, adjacency entries and fp32 features as bf16 hi + lo pairs, fp32 accumulation).
//
// What round 3's kernel lost (profiles/r03y_sage_mfma_isa.md: 480 registers, ONE wave per SIMD): a wave walked its edge
// streams, converted and scattered them, read fragments and issued MFMAs one after the other, and nothing else was resident to
// fill the matrix pipe meanwhile.  A first two-wave version of the same stream walk was SLOWER (13.3 ms against 7.3:
// profiles/r04c_sage_v2_ablation.json — the branch-free scatter of 8 prefetched entries per stream and chunk is ~400 vector
// instructions on the step's critical path, 6.3 ms of the 13.3; the self-loop rows of the epilogue another 1.5).  So the work
// is split differently:
//
//   1. sage_bcm_pack_kernel ("block-chunk-major" repack, one pass over the cell rows, HBM-bound): every group of 64 cells gets
//      its in-window entries sorted by K chunk (32 window genes = 2 MFMA steps): the weight next to a dword holding the entry's
//      element offset inside the group's A image.  1.6 GB read, 1.6 GB written at 1M x 2000 x 10 %.  The result depends on the
//      GRAPH only (not on alpha / the features): it is a plan the caller may keep for as long as rowptr / col / w are unchanged
//      (dh_sage_window_plan + dh_sage_window_mfma_planned; the model calls the aggregation three times per epoch on one graph).
//   2. sage_bcm_kernel: workgroup = 8 wavefronts = 2 cell groups of 64 x 4 column waves; a wave holds 64 cells x at most 4
//      column tiles of 32 (128 accumulator registers), <= 256 registers -> two waves per SIMD.  The 13 tiles of D = 400 are
//      dealt 4-3-3-3 to the waves of group 0 and 3-4-3-3 to group 1, so the two waves of every SIMD hold at most 7 tiles
//      together: the matrix-pipe floor is round 3's, 42 MFMAs per step and SIMD.  Per chunk a lane moves ONE packed entry:
//      two ds_read_b32 from a staging area, two ds_write_b16 into the double-buffered A image (cleared one step earlier, a
//      barrier apart).  No stream state, no searches, no conversions in the loop.
//   3. EVERY vector-memory operation of the loop is an LDS DMA — the feature blocks (K-major fragment order, 26 KB per step for
//      fp32 features; three-slot ring two steps ahead, 1 KB pieces dealt round-robin to the 8 waves) and the packed entries of
//      the chunk after next (4 bytes per lane and array).  No load targets a register, so the compiler has no register hazard
//      to guard with an s_waitcnt vmcnt(0) of its own and the hand-counted vmcnt is exact.
//   4. epilogue: the first out-of-window in-edge of a row (the self loop: the cell's OWN feature row, cold in every cache) is
//      fetched by LDS DMA for 64 rows at a time (all pieces in flight at once) and folded into the accumulators on their way
//      out; further out-of-window edges by a fix-up pass.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

#ifdef DH_SB_PROF  // development build: a cycle-stamped timeline (s_memtime) of every wave of ONE workgroup over a few steps
constexpr int TL_BLOCK = 3000, TL_J0 = 40, TL_STEPS = 6, TL_PROBES = 6;
__device__ unsigned long long dh_sb_timeline[8][TL_STEPS][TL_PROBES];
#define TLP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tl[k] = t_; } while (0)
#else
#define TLP(k)
#endif
#define PROF_T(x)
#ifndef DH_SB_SETPRIO
#define DH_SB_SETPRIO 1
#endif
#define PROF_ADD(i, a, b)

constexpr int JC = 2;                      // MFMA steps per chunk (32 window genes)
constexpr int IMG = 2 * JC * 2 * 64 * 8;   // bf16 elements of one group's A image: [2 planes][JC][2 halves][64 cells][8] = 8 KB
constexpr int PLANE = JC * 2 * 64 * 8;     // element offset of the lo plane
constexpr int MAX_CHUNKS = 128;            // n_cols <= 4096
constexpr int PACK_CAP = 16384;            // entries of a 64-cell group the pack kernel sorts in LDS (128 KB); larger groups go direct

__device__ __forceinline__ unsigned int f32_to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float widen(unsigned int h) { return __uint_as_float(h << 16); }

// HsP[j][plane][half][Dp][8] bf16 (step_elems apart): k slot s of half h of MFMA step j = window gene 16 j + 8 h + s
template <bool HBF16>
__global__ __launch_bounds__(256) void sage_bcm_prep_kernel(int64_t n_cols, int64_t width, int J, int Dp, const void* __restrict__ Hv,
                                                            int64_t ldh, uint16_t* __restrict__ HsP, int64_t step_elems) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (j, h, c)
  if (i >= (int64_t)J * 2 * Dp) return;
  const int c = (int)(i % Dp), h = (int)((i / Dp) % 2), j = (int)(i / (2 * Dp));
  uint16_t hi[8], lo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int64_t g = (int64_t)16 * j + 8 * h + s;
    const bool live = g < n_cols && c < width;
    if (HBF16) {
      hi[s] = live ? static_cast<const uint16_t*>(Hv)[g * ldh + c] : (uint16_t)0;
      lo[s] = 0;
    } else {
      const float v = live ? static_cast<const float*>(Hv)[g * ldh + c] : 0.f;
      const unsigned int a = f32_to_bf16(v);
      hi[s] = (uint16_t)a;
      lo[s] = (uint16_t)f32_to_bf16(v - widen(a));
    }
  }
  uint16_t* o = HsP + j * step_elems + (((int64_t)0 * 2 + h) * Dp + c) * 8;
#pragma unroll
  for (int s = 0; s < 8; ++s) o[s] = hi[s];
  if (!HBF16) {
    uint16_t* o1 = HsP + j * step_elems + (((int64_t)1 * 2 + h) * Dp + c) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) o1[s] = lo[s];
  }
}

// bounds[row][s] (s = 0 .. S) = the first stored entry of `row` whose column is >= col_begin + s * slice_cols, inside the row's
// in-window run (the out-of-window entries — self loops — sit at the row's ends and are skipped first); bounds[row][S] = the
// run's end.  One thread per (row, s): the plan of a sliced window finds every (row, slice) piece without walking the rows.
__global__ __launch_bounds__(256) void sage_bcm_bounds_kernel(int64_t n_dst, int col_begin, int n_cols, int slice_cols, int S,
                                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              int32_t* __restrict__ bounds) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_dst * (S + 1)) return;
  const int64_t row = i / (S + 1);
  const int sidx = (int)(i - row * (S + 1));
  int lo = rowptr[row], hi = rowptr[row + 1];
  while (lo < hi && (unsigned)(col[lo] - col_begin) >= (unsigned)n_cols) ++lo;
  while (hi > lo && (unsigned)(col[hi - 1] - col_begin) >= (unsigned)n_cols) --hi;
  if (sidx < S) {
    const int target = sidx * slice_cols;  // window position
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (col[mid] - col_begin < target) lo = mid + 1; else hi = mid;
    }
  } else {
    lo = hi;
  }
  bounds[i] = lo;
}

// One workgroup (16 wavefronts) per group of 64 destination rows (and, SLICED, per window slice): the group's entries with a
// column inside the slice, sorted by chunk (position in the slice / 32; order inside a chunk is whatever the LDS atomics give —
// every entry owns its own slot of the A image, so the product does not depend on it), land at pent[base ...) with
// base = rowptr[first row] + (the entries of the group's rows that precede the slice), and chunk_ptr[(group, slice)][c] points at
// chunk c's first entry (the entry behind the last chunk = one past the end).  An entry = {element offset inside the group's A
// image (plane 0), the weight's bits}, 8 bytes; the window position of an entry of chunk c is 32 c + 8 (offset >> 9) + (offset & 7).
//   Fast path (every row <= 256 entries in the slice, <= 16384 in the group — 10 % of 2000 columns is 200 +- 13): a wavefront owns
//   4 rows, a lane the entries lane, lane + 64, ... of each — 16 (column, weight) pairs per lane requested at once and kept in
//   registers across histogram -> scan -> placement into an LDS image of the sorted list -> one coalesced write-out.  The edge
//   list is read once.  Anything else takes the two-pass path with direct stores.
constexpr int PACK_ROWS = 4, PACK_IT = 4;  // rows per wavefront, 64-entry strides per row held in registers
template <bool SLICED>
__global__ __launch_bounds__(1024) void sage_bcm_pack_kernel(int64_t n_dst, int col_begin, int n_cols, int slice_cols, int S,
                                                             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             const float* __restrict__ w, const int32_t* __restrict__ bounds,
                                                             int32_t* __restrict__ chunk_ptr, u32x2* __restrict__ pent) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* const rlo = reinterpret_cast<int*>(smem);       // [64] the rows' first entry in the slice
  int* const rhi = rlo + 64;                           // [64] ... and one past their last
  int* const hist = rhi + 64;                          // [MAX_CHUNKS + 1]: counts, then exclusive offsets
  int* const cursor = hist + MAX_CHUNKS + 4;           // [MAX_CHUNKS]
  int* const misc = cursor + MAX_CHUNKS;               // [4]: {a row too long for the register path, entries before the slice, entries in it}
  u32x2* const s_ent = reinterpret_cast<u32x2*>(misc + 4);  // [PACK_CAP] {offset in the A image, weight bits}
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sl = SLICED ? (int)blockIdx.y : 0;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const int wbeg = col_begin + sl * slice_cols;                  // first column of the slice
  const int wn = min(slice_cols, n_cols - sl * slice_cols);      // its width
  const int n_chunks = (wn + 31) >> 5, cps1 = (slice_cols >> 5) + 1;
  for (int i = tid; i < MAX_CHUNKS + 1; i += 1024) hist[i] = 0;
  for (int i = tid; i < MAX_CHUNKS; i += 1024) cursor[i] = 0;
  if (tid < 4) misc[tid] = 0;
  __syncthreads();
  if (tid < 64) {
    const int64_t row = c0 + tid;
    int lo = 0, hi = 0, before = 0;
    if (row < n_dst) {
      const int rs = rowptr[row];
      if (SLICED) {
        lo = bounds[row * (S + 1) + sl];
        hi = bounds[row * (S + 1) + sl + 1];
      } else {
        lo = rs;
        hi = rowptr[row + 1];
      }
      before = lo - rs;
    }
    rlo[tid] = lo;
    rhi[tid] = hi;
    if (hi - lo > 64 * PACK_IT) misc[0] = 1;
    atomicAdd(&misc[1], before);
    atomicAdd(&misc[2], hi - lo);
  }
  __syncthreads();
  const int base = rowptr[min(c0, n_dst)] + misc[1];
  const bool fast = misc[0] == 0 && misc[2] <= PACK_CAP;  // block-uniform

  auto scan_and_publish = [&]() {
    __syncthreads();
    if (tid < 64) {  // exclusive scan of up to 128 counts by one wavefront: two per lane
      const int a = hist[2 * tid], b = hist[2 * tid + 1];
      int incl = a + b;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (tid >= d) incl += t;
      }
      const int excl = incl - (a + b);
      hist[2 * tid] = excl;
      hist[2 * tid + 1] = excl + a;
      if (tid == 63) hist[MAX_CHUNKS] = incl;
    }
    __syncthreads();
    const int total = hist[MAX_CHUNKS];
    int32_t* cp = chunk_ptr + ((int64_t)blockIdx.x * S + sl) * cps1;
    for (int i = tid; i < cps1; i += 1024) cp[i] = base + (i < n_chunks ? hist[i] : total);
    return total;
  };
  auto pack = [&](int g, float wv_e, int row, uint32_t& idx, uint32_t& hl) {
    const int gi = g & 31;
    idx = (uint32_t)(((((gi >> 4) * 2 + ((gi >> 3) & 1)) * 64 + row) * 8) + (gi & 7));
    hl = __float_as_uint(wv_e);
  };

  if (fast) {
    int gq[PACK_ROWS][PACK_IT];
    float wq[PACK_ROWS][PACK_IT];
    // Every lane loads from a valid index (lanes past a row's end: the group's first entry) and the value is dropped by a select:
    // written as `ok ? col[e] : -1` each of the 16 pairs sat in its own exec-masked block with a wait behind it (round 5's load
    // audit) — 16 dependent round trips where "16 pairs at once" was meant.  A group without entries loads nothing.
    const bool any = misc[2] > 0;  // block-uniform
    const int e_safe = any ? rowptr[min(c0, n_dst - 1)] : 0;
#pragma unroll
    for (int rr = 0; rr < PACK_ROWS; ++rr) {
      const int rs = rlo[PACK_ROWS * wv + rr], re = rhi[PACK_ROWS * wv + rr];
#pragma unroll
      for (int it = 0; it < PACK_IT; ++it) {
        const int e = rs + lane + 64 * it;
        const bool ok = e < re;
        int cv = 0;
        float wvv = 0.f;
        if (any) {
          const int ec = ok ? e : e_safe;
          cv = col[ec];
          wvv = w[ec];
        }
        gq[rr][it] = ok ? cv - wbeg : -1;
        wq[rr][it] = ok ? wvv : 0.f;
      }
    }
#pragma unroll
    for (int rr = 0; rr < PACK_ROWS; ++rr)
#pragma unroll
      for (int it = 0; it < PACK_IT; ++it)
        if ((unsigned)gq[rr][it] < (unsigned)wn) atomicAdd(&hist[gq[rr][it] >> 5], 1);
    const int total = scan_and_publish();
#pragma unroll
    for (int rr = 0; rr < PACK_ROWS; ++rr)
#pragma unroll
      for (int it = 0; it < PACK_IT; ++it) {
        const int g = gq[rr][it];
        if ((unsigned)g >= (unsigned)wn) continue;
        uint32_t idx, hl;
        pack(g, wq[rr][it], PACK_ROWS * wv + rr, idx, hl);
        const int pos = hist[g >> 5] + atomicAdd(&cursor[g >> 5], 1);
        s_ent[pos] = u32x2{idx, hl};
      }
    __syncthreads();
    for (int i = tid; i < total; i += 1024) pent[(int64_t)base + i] = s_ent[i];
    return;
  }
  // general path: two passes over the rows' pieces, direct (scattered) stores
  for (int rr = 0; rr < 64; ++rr)
    for (int e = rlo[rr] + tid; e < rhi[rr]; e += 1024) {
      const int g = col[e] - wbeg;
      if ((unsigned)g < (unsigned)wn) atomicAdd(&hist[g >> 5], 1);
    }
  scan_and_publish();
  for (int rr = 0; rr < 64; ++rr)
    for (int e = rlo[rr] + tid; e < rhi[rr]; e += 1024) {
      const int g = col[e] - wbeg;
      if ((unsigned)g >= (unsigned)wn) continue;
      uint32_t idx, hl;
      pack(g, w[e], rr, idx, hl);
      const int pos = hist[g >> 5] + atomicAdd(&cursor[g >> 5], 1);
      pent[(int64_t)base + pos] = u32x2{idx, hl};
    }
}

// neigh[row, c] = 1 / deg(row) * ( rowscale[row] * sum over the sets of partial[set][row][c]  +  the out-of-window in-edges with the
// alpha rule of gnn.py:72-76 ), the shares summed in set order (deterministic)
template <bool HBF16, bool OBF16>
__global__ __launch_bounds__(256) void sage_bcm_reduce_kernel(int64_t n_dst, int64_t width, int Dp, int n_sets, int col_begin, int n_cols,
                                                              const float* __restrict__ partial, const float* __restrict__ rowscale,
                                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              const float* __restrict__ w, const void* __restrict__ Hraw, int64_t ldh,
                                                              const int32_t* __restrict__ src_id, const int32_t* __restrict__ dst_id,
                                                              const float* __restrict__ alpha, int n_genes, void* __restrict__ neigh, int64_t ldn) {
  const int64_t row = blockIdx.x;
  const int rs = rowptr[row], re = rowptr[row + 1];
  int s0 = rs, e0 = re;
  while (s0 < e0 && (unsigned)(col[s0] - col_begin) >= (unsigned)n_cols) ++s0;
  while (e0 > s0 && (unsigned)(col[e0 - 1] - col_begin) >= (unsigned)n_cols) --e0;
  const float inv = re > rs ? 1.f / (float)(re - rs) : 0.f;
  const float rsc = rowscale ? rowscale[row] : 1.f;
  const int did = dst_id[row];
  for (int64_t c = threadIdx.x; c < width; c += 256) {
    float v = 0.f;
    {
      int s = 0;
      for (; s + 8 <= n_sets; s += 8) {  // eight partial values in flight, added in order (a plain loop is n_sets dependent round trips)
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = partial[((int64_t)(s + u) * n_dst + row) * Dp + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t8[u];
      }
      for (; s < n_sets; ++s) v += partial[((int64_t)s * n_dst + row) * Dp + c];
    }
    v *= rsc;
    for (int part = 0; part < 2; ++part)  // the out-of-window entries: [rs, s0) and [e0, re)
      for (int e = part ? e0 : rs; e < (part ? re : s0); ++e) {
        const int u = col[e];
        const int sid = src_id[u];
        int idx = n_genes + 1;
        if (sid >= 0 && did < 0) idx = sid;
        if (did >= 0 && sid < 0) idx = did;
        if (did >= 0 && sid >= 0) idx = n_genes;
        const float hv = HBF16 ? widen(static_cast<const uint16_t*>(Hraw)[(int64_t)u * ldh + c]) : static_cast<const float*>(Hraw)[(int64_t)u * ldh + c];
        v = fmaf(w[e] * alpha[idx], hv, v);
      }
    v *= inv;
    if (OBF16) static_cast<uint16_t*>(neigh)[row * ldn + c] = (uint16_t)f32_to_bf16(v);
    else static_cast<float*>(neigh)[row * ldn + c] = v;
  }
}

// NT = ceil(tiles / 4): a wave holds NT or NT - 1 column tiles (wave-uniform; the step loop is instantiated for both)
// SPLITK (gene <- cell: few destination rows, a window of up to millions of columns): the window is cut into slices of `slice_cols`
// = 2048 columns (64 chunks; the plan holds the packed entries and chunk pointers per 64-row group AND slice, a group's slices one
// behind the other), set s (see set_id) owns `slices_per_set` consecutive slices, runs the loop over them as ONE long K range and
// writes its fp32 share to partial[set][row][Dp]; sage_bcm_reduce_kernel sums the shares in order and adds the row scale, the mean
// and the out-of-window edges.  No column scale in this mode (gnn.py:74: a cell -> gene edge takes alpha of the DESTINATION gene).
// The feature blocks of a set are read by its 16 row blocks in near lockstep: the first one's loads bring a block from HBM, the
// others find it in the XCD's L2 (the set -> XCD mapping below).  Touching blocks ahead of time and walking the blocks in a different
// order per row block (to spread simultaneous requests over the L2 channels) were both tried and both cost 3 - 6 %
// (profiles/r04y_sage_splitk_ab.json): the one-step lead of the movers is enough.  What does matter: every LDS region behind the
// chunk-pointer table must start on a 16-byte boundary — with an 8-byte offset every 16-byte LDS access of the loop is split and
// the step takes 8500 cycles instead of 2800.
template <bool HBF16, bool OBF16, int NT, bool SPLITK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void sage_bcm_kernel(
    int64_t n_dst, int64_t width, int col_begin, int n_cols, int slice_cols, int n_slices, int slices_per_set, int cptr_stride, int Dp,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ colscale,
    const int32_t* __restrict__ chunk_ptr, const u32x2* __restrict__ pent, const uint16_t* __restrict__ HsP, void* __restrict__ neigh,
    int64_t ldn, int64_t nnz, int step_bytes, const void* __restrict__ Hraw, int64_t ldh, const int32_t* __restrict__ src_id,
    const int32_t* __restrict__ dst_id, const float* __restrict__ alpha, int n_genes, float* __restrict__ partial) {
  constexpr int P = HBF16 ? 1 : 2;
  const int cps1 = (slice_cols >> 5) + 1;  // chunk pointers per group and slice in the plan
  constexpr int HS = HBF16 ? 2 : 4;  // bytes per feature
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // persistent through the epilogue: rowinfo, ends.  Loop: chunk pointers, staging, A images, B ring; epilogue: the row buffer over them.
  f32x4_t* const rowinfo = reinterpret_cast<f32x4_t*>(smem);                 // [128] {1 / deg, f0, bits(u0), bits(n_tail)}
  int* const ends = reinterpret_cast<int*>(rowinfo + 128);                   // [128][2]: in-window run [s0, e0)
  unsigned char* const scratch = reinterpret_cast<unsigned char*>(ends + 256);
  int* const cptr = reinterpret_cast<int*>(scratch);                         // [2 groups][cptr_stride]: the chunk pointers of this workgroup's slices
  float* const cs = reinterpret_cast<float*>(cptr + 2 * cptr_stride);        // [slice_cols]: colscale of the window columns (not SPLITK)
  uint16_t* const a_img = reinterpret_cast<uint16_t*>(cs + (SPLITK ? 0 : slice_cols));  // [2 buffers][2 groups][IMG]
  unsigned char* const b_img = reinterpret_cast<unsigned char*>(a_img + 2 * 2 * IMG);  // [3 slots][step_bytes]
  unsigned char* const rowbuf = scratch;                                     // epilogue: [64 rows][width * HS] (16-byte pieces, row-linear)

  PROF_T(t_start);
  // SPLITK: a 1-D grid whose workgroups i, i + 8, i + 16, ... share an XCD (and its L2): the row blocks of one set are neighbours
  // THERE, so the set's feature blocks come out of HBM once and are read by its row blocks from L2
  const int row_blocks = (int)((n_dst + 127) / 128);
  const int rb_id = SPLITK ? (int)((blockIdx.x >> 3) % row_blocks) : (int)blockIdx.x;
  const int set_id = SPLITK ? (int)((blockIdx.x >> 3) / row_blocks) * 8 + (int)(blockIdx.x & 7) : 0;
  if (SPLITK && set_id * slices_per_set >= n_slices) return;  // (the grid rounds the sets up to a multiple of 8)
  // this workgroup's part of the window: slices [sl0, sl1) = columns [col0, col0 + wn) = J MFMA steps = n_chunks chunks.  A slice has
  // its own chunk-pointer row in the plan (cps1 entries, the last one = the slice's end), so chunk c's pointer sits at c + c / (cps1 - 1)
  const int sl0 = set_id * slices_per_set, sl1 = SPLITK ? min(n_slices, sl0 + slices_per_set) : 1;
  const int wn = SPLITK ? min(n_cols, sl1 * slice_cols) - sl0 * slice_cols : n_cols;
  const int J = (wn + 15) / 16, n_chunks = (J + JC - 1) / JC;
  const unsigned char* const hsp = reinterpret_cast<const unsigned char*>(HsP) + (int64_t)sl0 * (slice_cols / 16) * step_bytes;
  auto ci = [&](int c) __attribute__((always_inline)) { return SPLITK ? c + (c >> 6) : c; };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int r = lane & 31, half = lane >> 5;
  const int T = Dp / 32, tbase = T / 4, trem = T % 4;
  auto extra = [&](int q) { return (((q - grp * trem) % 4 + 4) % 4) < trem ? 1 : 0; };
  const int my_tiles = tbase + extra(wq);
  int tile0 = wq * tbase;
  for (int q = 0; q < wq; ++q) tile0 += extra(q);
  const int64_t cell0 = (int64_t)rb_id * 128 + grp * 64;  // first cell of the group
  const int li = wq * 64 + lane;                               // this lane's entry slot inside a chunk's list (0 .. 255)

  // ---- prologue -----------------------------------------------------------------------------------------------------------
  const int64_t grp_id = (int64_t)rb_id * 2 + grp;
  const bool live_grp = grp_id * 64 < n_dst;
  for (int i = li; i < (sl1 - sl0) * cps1; i += 256)  // the chunk pointers of (group, slices sl0 .. sl1 - 1)
    cptr[grp * cptr_stride + i] = live_grp ? chunk_ptr[(grp_id * n_slices + sl0) * cps1 + i] : 0;
  if (!SPLITK)
    for (int i = tid; i < wn; i += 512) cs[i] = colscale ? colscale[i] : 1.f;
  if (!SPLITK && wq == 0) {  // per-row table of the group's 64 cells
    const int cl = grp * 64 + lane;
    const int64_t my_cell = cell0 + lane;
    int s0 = 0, e0 = 0;
    f32x4_t ri = {0.f, 0.f, 0.f, 0.f};
    if (my_cell < n_dst) {
      const int rs = rowptr[my_cell], re = rowptr[my_cell + 1];
      s0 = rs;
      e0 = re;
      while (s0 < e0 && (unsigned)(col[s0] - col_begin) >= (unsigned)n_cols) ++s0;      // out-of-window edges sit at the ends
      while (e0 > s0 && (unsigned)(col[e0 - 1] - col_begin) >= (unsigned)n_cols) --e0;
      const int d = re - rs;
      const int n_tail = (s0 - rs) + (re - e0);
      ri[0] = d > 0 ? 1.f / (float)d : 0.f;
      if (n_tail > 0) {  // the first out-of-window in-edge (the self loop): alpha rule of gnn.py:72-76
        const int e = s0 > rs ? rs : e0;
        const int u = col[e];
        const int sid = src_id[u], did = dst_id[my_cell];
        int idx = n_genes + 1;
        if (sid >= 0 && did < 0) idx = sid;
        if (did >= 0 && sid < 0) idx = did;
        if (did >= 0 && sid >= 0) idx = n_genes;
        ri[1] = w[e] * alpha[idx];
        ri[2] = __int_as_float(u);
      }
      ri[3] = __int_as_float(n_tail);
    }
    ends[2 * cl] = s0;
    ends[2 * cl + 1] = e0;
    rowinfo[cl] = ri;
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][y][i] = 0.f;

  // ---- loop machinery --------------------------------------------------------------------------------------------------------
  // Everything the loop fetches from memory goes through REGISTERS (plain loads, then LDS stores): an LDS-DMA instruction costs a
  // wave 100 - 300 cycles of issue time next to MFMAs and LDS reads (measured: 1050 cycles per step and wave for 3 - 4 pieces,
  // profiles/r04g_sage_bcm_phases.json — more than its 24 MFMAs), a global_load + ds_write_b128 pair ~25.
  //   vmcnt retires in order, so a wave that has a packed-entry load (HBM: 1 - 2 us) in its queue cannot wait for a younger feature
  // load (L2: ~0.4 us) without waiting for the entry first; with both kinds in every wave each step stalled on HBM latency however
  // the loads were ordered (profiles/r04i_sage_bcm_phases.json).  So the waves SPECIALISE: of a group's four column waves the last
  // two (three tiles each at D = 400, i.e. the ones with MFMA time to spare) move the feature blocks for the whole workgroup, the
  // first two load and scatter the group's packed entries.  Each queue then holds one kind of load, and whatever wait the compiler
  // puts in front of a use is the right one.
  //   * movers: feature block of step j + 2, 1 KB pieces, piece k belongs to mover k % 4; loaded at the END of step j, written to
  //     LDS slot (j + 2) % 3 at the end of step j + 1 (that slot's block j - 1 was last read by the lagging group early in interval
  //     j, a barrier ago), read from interval j + 2 on.  Every mover moves exactly NPM pieces, unconditionally: a k beyond the block
  //     re-reads the last piece and writes it into the pad behind the slot's data.
  //   * entry waves: lane l of the pair owns entries l and l + 128 of every chunk's list; they are loaded right after the previous
  //     chunk's were scattered (two steps ahead of their own scatter) and scattered straight from the registers.
  const bool mover = wq >= 2;
  const int mrank = grp * 2 + (wq & 1);              // among the four movers
  constexpr int NPM = HBF16 ? NT : 2 * NT;          // ceil(pieces / 4) for every width of this instantiation (pieces = P Dp / 32 <= 4 P NT)
  constexpr int SLOT_KB = NPM * 4;                  // LDS bytes of a feature slot, in KB
  const int pieces = step_bytes >> 10;
  u32x4 rb[NPM];
  const uint32_t voff = (uint32_t)lane * 16;
  auto b_load_part = [&](int j, auto lo_c, auto hi_c) __attribute__((always_inline)) {  // pieces [LO, HI) of this mover's NPM
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    const unsigned char* blk = hsp + (int64_t)j * step_bytes;
#pragma unroll
    for (int s = LO; s < HI; ++s) {
      const unsigned char* base = blk + (size_t)min(mrank + 4 * s, pieces - 1) * 1024;  // wave-uniform: scalar base + lane offset
      rb[s] = *reinterpret_cast<const u32x4*>(base + voff);
    }
  };
  auto b_load = [&](int j) __attribute__((always_inline)) { b_load_part(j, std::integral_constant<int, 0>{}, std::integral_constant<int, NPM>{}); };
  auto b_store = [&](int j) __attribute__((always_inline)) {
    unsigned char* slot = b_img + (size_t)(j % 3) * (SLOT_KB * 1024);
#pragma unroll
    for (int s = 0; s < NPM; ++s) *reinterpret_cast<u32x4*>(slot + (size_t)(mrank + 4 * s) * 1024 + voff) = rb[s];
  };
  const int eli = (wq & 1) * 64 + lane;  // entry waves: this lane's first entry slot (the second is eli + 128)
  u32x2 ent[2] = {{0u, 0u}, {0u, 0u}};
  auto ent_load = [&](int c) __attribute__((always_inline)) {  // (lanes beyond the list read a neighbour, unused)
    const int64_t e = (int64_t)cptr[grp * cptr_stride + ci(c)] + eli;
    ent[0] = __builtin_nontemporal_load(pent + min(e, nnz - 1));
    ent[1] = __builtin_nontemporal_load(pent + min(e + 128, nnz - 1));
  };
  // this lane's entries of chunk c (in `ent`) -> A buffer c & 1 (cleared one step earlier): a = w * colscale[gene] split into bf16
  // hi + lo (the residual 2^-18 |a|), two 2-byte stores each
  // round 6: each of the lane's two entries went through its own exec-masked block — scale look-up in LDS, a full lgkmcnt(0) wait, two
  // bf16 conversions compiled as branches around their NaN case, two stores — one after the other.  Now both look-ups are issued
  // together (clamped index: valid for every lane), the conversions are selects, and only the stores are predicated: 3.66 -> 3.55 ms
  // bf16, 5.11 -> 5.02 fp32 (A/B on one box).  Two further re-orderings were built and measured and are NOT here: the leading group's
  // scatter at the START of its step (under the partner's MFMA burst) and the scatter split into a request half in front of the wave's
  // MFMAs and a store half behind them, entries requested two steps ahead — both 0 % (fp32) / +4 % (bf16) against this form: with the
  // probes' own lgkmcnt waits removed, the scatter is not what the step waits for.
  auto bf16_rne = [](float x) __attribute__((always_inline)) {  // f32_to_bf16 as a select (same bits)
    const unsigned int u = __float_as_uint(x);
    const unsigned int rounded = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16, quiet = (u >> 16) | 0x40u;
    return (u & 0x7fffffffu) > 0x7f800000u ? quiet : rounded;
  };
  auto put2 = [&](uint16_t* img, int c, const u32x2 (&en)[2], bool ok0, bool ok1) __attribute__((always_inline)) {
    float a[2] = {__uint_as_float(en[0][1]), __uint_as_float(en[1][1])};
    if (!SPLITK) {
      float sc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) sc[q] = cs[min(32 * c + (int)(en[q][0] >> 9) * 8 + (int)(en[q][0] & 7), wn - 1)];
#pragma unroll
      for (int q = 0; q < 2; ++q) a[q] *= sc[q];
    }
    unsigned int ahi[2], alo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ahi[q] = bf16_rne(a[q]);
      alo[q] = bf16_rne(a[q] - widen(ahi[q]));
    }
    if (ok0) {
      img[en[0][0]] = (uint16_t)ahi[0];
      img[en[0][0] + PLANE] = (uint16_t)alo[0];
    }
    if (ok1) {
      img[en[1][0]] = (uint16_t)ahi[1];
      img[en[1][0] + PLANE] = (uint16_t)alo[1];
    }
  };
  auto put = [&](uint16_t* img, int c, u32x2 en) __attribute__((always_inline)) {  // the rare rounds beyond 256 entries
    float a = __uint_as_float(en[1]);
    if (!SPLITK) a *= cs[min(32 * c + (int)(en[0] >> 9) * 8 + (int)(en[0] & 7), wn - 1)];
    const unsigned int ahi = f32_to_bf16(a);
    const unsigned int alo = f32_to_bf16(a - widen(ahi));
    img[en[0]] = (uint16_t)ahi;
    img[en[0] + PLANE] = (uint16_t)alo;
  };
  auto scatter = [&](int c) __attribute__((always_inline)) {
    uint16_t* const img = a_img + (size_t)((c & 1) * 2 + grp) * IMG;
    const int first = cptr[grp * cptr_stride + ci(c)];
    const int len = cptr[grp * cptr_stride + ci(c) + 1] - first;
    // (lanes beyond the list hold a neighbour's entry: a valid offset and a finite weight, computed and dropped)
    put2(img, c, ent, eli < len, eli + 128 < len);
    // a group with more than 256 entries in one chunk (mean 205 at 10 % density, ~4 sigma): the rest in rounds, loaded on the spot
    for (int rnd = 256; __builtin_expect(rnd < len, 0); rnd += 128) {
      const u32x2 en = pent[min((int64_t)first + rnd + eli, nnz - 1)];
      if (rnd + eli < len) put(img, c, en);
    }
  };
  auto clear = [&](int c) __attribute__((always_inline)) {  // the group's 256 lanes zero the 8 KB of buffer c & 1
    u32x4* const z = reinterpret_cast<u32x4*>(a_img + (size_t)((c & 1) * 2 + grp) * IMG);
    z[li] = u32x4(0u);
    z[li + 256] = u32x4(0u);
  };

  bf16x8_t fa[2][2];
  auto read_a = [&](int j) __attribute__((always_inline)) {  // the A fragments of step j: chunk j / JC -> buffer, j % JC -> slot
    const uint16_t* img = a_img + (size_t)(((j / JC) & 1) * 2 + grp) * IMG;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        fa[pl][m] = *reinterpret_cast<const bf16x8_t*>(img + ((size_t)((pl * JC + (j % JC)) * 2 + half) * 64 + 32 * m + r) * 8);
  };

  // Chunk c = MFMA steps 2c, 2c + 1; the A buffers alternate with the chunk, the feature slots with the step.
  //   The two waves of a SIMD belong to different groups, and the groups run the step out of phase (LAG): group 0 does its
  // bookkeeping (clear / scatter, loads, stores, B-fragment reads) and THEN its MFMAs of step j; group 1 reads the fragments of step
  // j at the end of interval j and issues those MFMAs FIRST THING in interval j + 1 — while its SIMD partner is still in the
  // bookkeeping — so that the matrix pipe has work from the barrier on.
  //   An A buffer may be rewritten once the group's last fragment read of its old chunk is behind a barrier: the lagging group reads
  // step j's fragments IN interval j, the leading group at the end of interval j - 1 — so the leading group clears and fills one step
  // earlier than the lagging one, and both leave a whole interval between the scatter and the first read:
  //   lagging: even step 2c: clear(c + 1);        odd step 2c + 1: scatter(c + 1), then load the entry of chunk c + 2
  //   leading: even step 2c: scatter(c + 1), then load the entry of chunk c + 2;   odd step 2c + 1: clear(c + 2)
  // Feature slot (j + 1) % 3 is free at the end of step j (its block j - 2 was last read by the lagging group's MFMAs at the start of
  // interval j - 1): block j + 1 goes there.
  auto run = [&](auto mt_c, auto lag_c) __attribute__((always_inline)) {
    constexpr int MT = decltype(mt_c)::value;
    constexpr bool LAG = decltype(lag_c)::value != 0;
    // B fragments just in time: tile y + 1 is read while tile y's 4 or 6 MFMAs issue (two fragment sets: 8 or 16 registers
    // instead of all tiles' 16 or 32 — with them resident the 4-tile fp32 instantiation spilled accumulators around every MFMA block)
    bf16x8_t fb[2][P];
    auto read_fb = [&](int j, int y, int set) __attribute__((always_inline)) {
      const unsigned char* bb = b_img + (size_t)(j % 3) * (SLOT_KB * 1024);
#pragma unroll
      for (int pl = 0; pl < P; ++pl)
        fb[set][pl] = *reinterpret_cast<const bf16x8_t*>(bb + ((size_t)(pl * 2 + half) * Dp + (tile0 + y) * 32 + r) * 16);
    };
    auto mma_tile = [&](auto y_c) __attribute__((always_inline)) {
      constexpr int y = decltype(y_c)::value;
#ifdef DH_SB_NO_MFMA
      return;
#endif
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[y & 1][0], acc[m][y], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][m], fb[y & 1][0], acc[m][y], 0, 0, 0);
      if (!HBF16) {
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[y & 1][P - 1], acc[m][y], 0, 0, 0);
      }
    };
    // the MFMAs of step j; tile 0's fragments (and the A fragments) are already in registers.  `hook(y)` runs after tile y's MFMAs
    // are issued (the leading group's entry waves put their entry request behind tile 0)
    auto mma = [&](int j, auto&& hook) __attribute__((always_inline)) {
      // the wave inside its MFMA block goes ahead of its SIMD partner's bookkeeping (scatter arithmetic, LDS traffic) at issue:
      // 4.87 -> 4.76 ms fp32 with priority 1 (3: 4.79; bf16 unchanged), A/B builds on one box, round 6
      __builtin_amdgcn_s_setprio(DH_SB_SETPRIO);
      static_for<MT>([&](auto y_c) __attribute__((always_inline)) {
        constexpr int y = decltype(y_c)::value;
        if constexpr (y + 1 < MT) read_fb(j, y + 1, (y + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tile(y_c);
        __builtin_amdgcn_sched_barrier(0);
        hook(y_c);
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_s_setprio(0);
    };
    // One step.  PAR = its parity (selects the bookkeeping role), FULL = steady state: every block / chunk it touches exists, so all
    // of its loads and stores are unconditional.
    //   Who touches memory when — the address unit takes 25 - 60 cycles per 1 KB load instruction and a wave blocks on issue while it
    //   is busy: 28 loads issued together by the four movers cost each of them, and any other wave loading at that time, 700 - 1300
    //   cycles (profiles/r04m_sage_bcm_timeline.txt).  So: the leading group's movers store and load at the START of the step (their
    //   group is in its bookkeeping then, their SIMD partners in their MFMAs), the lagging group's movers at the END (after their
    //   MFMAs and fragment reads); the entry waves request the next entries in between — in the step after their scatter, the
    //   lagging ones right behind their MFMAs, the leading ones behind their first tile's — and scatter at the END of the following
    //   step (one and a half steps of flight; the scattered chunk is first read a whole step later).  (Spreading a mover's loads
    //   over the tiles of its MFMA block instead was slower: every load stalls the wave's MFMA issue, 5.9 ms against 5.0.)
    auto step = [&](auto par_c, auto full_c, int j) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr bool FULL = decltype(full_c)::value != 0;
      constexpr bool FILL = LAG == (PAR == 1);  // the step that scatters: lagging group odd, leading group even
#ifdef DH_SB_PROF
      unsigned long long tl[TL_PROBES] = {0, 0, 0, 0, 0, 0};
#endif
      const int c = j / JC;
      // the entry request of the step after a scatter: the lagging group scatters chunk k at the end of step 2k - 1, the leading
      // group at the end of step 2k - 2; the request for chunk k + 1 follows one step later (step 2k: c + 1, resp. step 2k - 1: c + 2)
      auto request = [&]() __attribute__((always_inline)) {
        if (!mover && !FILL) {
          if (LAG) {
            if (FULL || c + 1 < n_chunks) ent_load(c + 1);
          } else if (FULL || c + 2 < n_chunks) ent_load(c + 2);
        }
      };
      auto lead_hook = [&](auto y_c) __attribute__((always_inline)) {
        if constexpr (decltype(y_c)::value == 0) request();
      };
      TLP(0);  // after the barrier
      if (!LAG && mover) {
        if (FULL || j + 1 < J) b_store(j + 1);  // loaded at the start of step j - 1 (block 1: in the prologue)
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || j + 2 < J) b_load(j + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LAG && (FULL || j > 0)) {  // step j - 1: its A fragments and its first tile's B fragments were read before the barrier
        __builtin_amdgcn_sched_barrier(0);
        mma(j - 1, [](auto) {});
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LAG) request();
      TLP(1);  // lagging group: its MFMAs are issued
      __builtin_amdgcn_sched_barrier(0);
      if (!FILL) {
        if (LAG) {
          if (FULL || c + 1 < n_chunks) clear(c + 1);
        } else if (FULL || c + 2 < n_chunks) clear(c + 2);
      }
      TLP(2);  // clear done
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MT > 0) {
        if (LAG) {
          read_a(j);
          read_fb(j, 0, 0);
        } else {
          read_fb(j, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          TLP(3);  // leading group: first B fragments in registers
          mma(j, lead_hook);
          TLP(4);  // leading group: its MFMAs are issued
          if (FULL || j + 1 < J) read_a(j + 1);
        }
      } else if (!LAG) request();
      __builtin_amdgcn_sched_barrier(0);
      if (FILL && !mover && (FULL || c + 1 < n_chunks)) scatter(c + 1);  // requested in the previous step
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise hoists the loads below above the MFMA block: their registers would be live across it)
      if (LAG && mover) {
        if (FULL || j + 1 < J) b_store(j + 1);  // loaded at the end of step j - 1 (block 1: in the prologue)
        __builtin_amdgcn_sched_barrier(0);
        if (FULL || j + 2 < J) b_load(j + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      TLP(5);  // fragments read, entries scattered, feature pieces stored and requested: at the barrier
      __syncthreads();
#ifdef DH_SB_PROF
      if (blockIdx.x == (SPLITK ? 100 : TL_BLOCK) && j >= (SPLITK ? 1000 : TL_J0) && j < (SPLITK ? 1000 : TL_J0) + TL_STEPS && lane == 0)
        for (int k = 0; k < TL_PROBES; ++k) dh_sb_timeline[wave][j - (SPLITK ? 1000 : TL_J0)][k] = tl[k];
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (!LAG && J > 0) read_a(0);
    if (!mover && n_chunks > 1) ent_load(1);  // scattered at the end of step 0 (leading) resp. step 1 (lagging)
    int j = 0;
    if (LAG) {  // the lagging group's step 0 has no MFMAs to start with: peel one chunk (the steady-state step has them unconditionally)
      if (J > 0) step(I0{}, I0{}, 0);
      if (J > 1) step(I1{}, I0{}, 1);
      j = 2;
    }
    for (; j + 5 < J; j += 2) {  // steady state: steps j, j + 1 with blocks up to j + 3 and chunks up to j / 2 + 2 (< n_chunks as 2 (c + 2) < J)
      step(I0{}, I1{}, j);
      step(I1{}, I1{}, j + 1);
    }
    for (; j < J; j += 2) {  // the last few steps, guarded
      step(I0{}, I0{}, j);
      if (j + 1 < J) step(I1{}, I0{}, j + 1);
    }
    if (LAG && J > 0) mma(J - 1, [](auto) {});
  };
  auto start_slice = [&]() __attribute__((always_inline)) {  // cleared images, chunk 0 scattered, feature block 0 in LDS, block 1 in registers
    clear(0);
    clear(1);
    __syncthreads();  // chunk pointers, column scales, (row table,) cleared images
    if (!mover && n_chunks > 0) {
      ent_load(0);
      scatter(0);
    }
    if (mover && J > 0) {  // block 0 straight into its slot; block 1 waits in the registers for step 0's store
      b_load(0);
      b_store(0);
    }
    if (mover && J > 1) b_load(1);
    __syncthreads();  // chunk 0 of both groups and feature block 0 are in place
  };
  auto run_slice = [&]() __attribute__((always_inline)) {
    if (grp == 0) {
      if (my_tiles == NT) run(std::integral_constant<int, NT>{}, std::integral_constant<int, 0>{});
      else run(std::integral_constant<int, (NT > 0 ? NT - 1 : 0)>{}, std::integral_constant<int, 0>{});
    } else {
      if (my_tiles == NT) run(std::integral_constant<int, NT>{}, std::integral_constant<int, 1>{});
      else run(std::integral_constant<int, (NT > 0 ? NT - 1 : 0)>{}, std::integral_constant<int, 1>{});
    }
  };
  start_slice();
  run_slice();
  if (SPLITK) {
    // this set's share of the window product, fp32, rows Dp wide; scales / mean / out-of-window edges are the reduce kernel's
    auto store_partial = [&](auto mt_c) __attribute__((always_inline)) {
      constexpr int MT = decltype(mt_c)::value;
      static_for<2 * (MT > 0 ? MT : 0)>([&](auto t_c) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int m = t / (MT > 0 ? MT : 1), y = t % (MT > 0 ? MT : 1);
        const int c = (tile0 + y) * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t cell = cell0 + 32 * m + (i & 3) + 8 * (i >> 2) + 4 * half;
          if (cell < n_dst) partial[((int64_t)set_id * n_dst + cell) * Dp + c] = acc[m][y][i];
        }
      });
    };
    if (my_tiles == NT) store_partial(std::integral_constant<int, NT>{});
    else store_partial(std::integral_constant<int, (NT > 0 ? NT - 1 : 0)>{});
    return;
  }

  PROF_T(t_epi);
  // ---- epilogue: neigh[cell, col] = (acc + f0 * H[u0, col]) / deg(cell).  C layout: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5).
  // For m = 0, 1: the self-loop rows of the 64 cells {group g, 32 m + r} are pulled into LDS by DMA — 16-byte pieces, every
  // piece of every row in flight at once — and folded into the accumulators on their way out.
  const int ppr = (int)(width * HS / 16);  // 16-byte pieces per row (the host checked divisibility and alignment)
  auto epilogue = [&](auto mt_c) __attribute__((always_inline)) {
    constexpr int MT = decltype(mt_c)::value;
    static_for<2>([&](auto m_c) __attribute__((always_inline)) {
      constexpr int m = decltype(m_c)::value;
      __syncthreads();  // the loop's LDS (m = 0) resp. the previous half's row buffer (m = 1) is dead
      {
        const int n_pieces = 64 * ppr;
        for (int q0 = 0; q0 < n_pieces; q0 += 512) {  // piece q of the buffer lands at q * 16: a wave's 64 pieces are contiguous
          const int q = q0 + tid;
          if (q < n_pieces) {  // (lanes beyond the last piece issue nothing: an active lane l writes LDS base + 16 l)
            const int row = q / ppr, pc = q - row * ppr;
            const f32x4_t ri = rowinfo[(row >> 5) * 64 + 32 * m + (row & 31)];
            const int64_t u = __float_as_int(ri[3]) > 0 ? (int64_t)__float_as_int(ri[2]) : 0;  // rows without such an edge read row 0 (unused)
            const unsigned char* src = static_cast<const unsigned char*>(Hraw) + (u * ldh) * HS + (size_t)pc * 16;
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(rowbuf + (size_t)(q0 + wave * 64) * 16), 16, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if constexpr (MT > 0) {
        static_for<MT>([&](auto y_c) __attribute__((always_inline)) {
          constexpr int y = decltype(y_c)::value;
          const int64_t c = (int64_t)(tile0 + y) * 32 + r;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * half;  // inside the 32-row tile
            const f32x4_t ri = rowinfo[grp * 64 + 32 * m + row];
            const int64_t cell = cell0 + 32 * m + row;
            if (cell < n_dst && c < width) {
              float hv = 0.f;
              if (__float_as_int(ri[3]) > 0) {
                const unsigned char* hp = rowbuf + (size_t)(grp * 32 + row) * ppr * 16 + (size_t)c * HS;
                hv = HBF16 ? widen(*reinterpret_cast<const uint16_t*>(hp)) : *reinterpret_cast<const float*>(hp);
              }
              const float v = fmaf(ri[1], hv, acc[m][y][i]) * ri[0];
              if (OBF16) static_cast<uint16_t*>(neigh)[cell * ldn + c] = (uint16_t)f32_to_bf16(v);
              else __builtin_nontemporal_store(v, static_cast<float*>(neigh) + cell * ldn + c);
            }
          }
        });
      }
    });
  };
  if (my_tiles == NT) epilogue(std::integral_constant<int, NT>{});
  else epilogue(std::integral_constant<int, (NT > 0 ? NT - 1 : 0)>{});

  // rows with more than one out-of-window in-edge (not a CellFeatureGraph row, but legal): the rest is added to the written
  // result element-wise (read-modify-write after the block's own stores; only this block touches these rows)
  __syncthreads();
  for (int rl = wave; rl < 128; rl += 8) {
    const f32x4_t ri = rowinfo[rl];
    const int n_tail = __float_as_int(ri[3]);
    if (n_tail < 2) continue;  // wave-uniform
    const int64_t cell = (int64_t)rb_id * 128 + rl;
    const int rs = rowptr[cell];
    const int s0 = ends[2 * rl], e0 = ends[2 * rl + 1];
    const int did = dst_id[cell];
    for (int q = 1; q < n_tail; ++q) {
      const int e1 = rs + q + (q < s0 - rs ? 0 : e0 - s0);
      const int u = col[e1];
      const int sid = src_id[u];
      int idx = n_genes + 1;
      if (sid >= 0 && did < 0) idx = sid;
      if (did >= 0 && sid < 0) idx = did;
      if (did >= 0 && sid >= 0) idx = n_genes;
      const float f = w[e1] * alpha[idx] * ri[0];
      for (int64_t c = lane; c < width; c += 64) {
        const float hvv = HBF16 ? widen(static_cast<const uint16_t*>(Hraw)[(int64_t)u * ldh + c]) : static_cast<const float*>(Hraw)[(int64_t)u * ldh + c];
        if (OBF16) {
          uint16_t* o = static_cast<uint16_t*>(neigh) + cell * ldn + c;
          *o = (uint16_t)f32_to_bf16(fmaf(f, hvv, widen(*o)));
        } else {
          float* o = static_cast<float*>(neigh) + cell * ldn + c;
          *o = fmaf(f, hvv, *o);
        }
      }
    }
  }
}

struct GeoB {
  int J, n_chunks, Dp, T, nt;
  int step_bytes;
  size_t prep_bytes, cptr_bytes, lds_bytes;
};
GeoB geometry_b(int64_t n_dst, int64_t n_cols, int64_t width, bool hbf16) {
  GeoB g;
  g.J = (int)((n_cols + 15) / 16);
  g.n_chunks = (g.J + JC - 1) / JC;
  g.Dp = (int)((width + 31) / 32 * 32);
  g.T = g.Dp / 32;
  g.nt = (g.T + 3) / 4;
  const int P = hbf16 ? 1 : 2;
  g.step_bytes = P * 2 * g.Dp * 16;  // a whole number of KB: Dp is a multiple of 32
  g.prep_bytes = ((size_t)g.J * g.step_bytes + 255) / 256 * 256;
  g.cptr_bytes = ((size_t)dh::ceil_div(n_dst, 64) * (g.n_chunks + 1) * 4 + 255) / 256 * 256;
  const int npm = hbf16 ? g.nt : 2 * g.nt;  // pieces per mover wave and step (the kernel's NPM)
  const size_t loop = (size_t)2 * (MAX_CHUNKS + 4) * 4 + (size_t)32 * g.n_chunks * 4 + (size_t)2 * 2 * IMG * 2 + (size_t)3 * npm * 4096;
  const size_t epi = (size_t)64 * width * (hbf16 ? 2 : 4) + 1024;
  g.lds_bytes = 128 * 16 + 256 * 4 + (loop > epi ? loop : epi);
  return g;
}
constexpr size_t PACK_LDS = (64 + 64 + MAX_CHUNKS + 4 + MAX_CHUNKS + 4) * 4 + (size_t)PACK_CAP * 8;

}  // namespace

#ifdef DH_SB_PROF
extern "C" __attribute__((visibility("default"))) int dh_sage_bcm_prof_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dh_sb_timeline), sizeof(unsigned long long) * 8 * TL_STEPS * TL_PROBES) != hipSuccess) return -1;
  return 0;
}
#endif

namespace dh {

bool sage_bcm_fits(int64_t n_dst, int64_t n_cols, int64_t width, bool hbf16, const void* H, int64_t ldh, int64_t nnz) {
  if (n_cols <= 0 || width <= 0 || width > 512 || n_cols > 32 * MAX_CHUNKS || nnz < 2 || nnz >= (int64_t)1 << 31) return false;
  const int hs = hbf16 ? 2 : 4;
  // the epilogue pulls whole feature rows into LDS in 16-byte pieces
  if ((width * hs) % 16 != 0 || (ldh * hs) % 16 != 0 || (reinterpret_cast<uintptr_t>(H) & 15u) != 0) return false;
  const GeoB g = geometry_b(n_dst, n_cols, width, hbf16);
  return g.lds_bytes <= 160 * 1024;
}

size_t sage_bcm_plan_bytes(int64_t n_dst, int64_t n_cols, int64_t nnz) {
  const GeoB g = geometry_b(n_dst, n_cols, 32, false);
  return g.cptr_bytes + (size_t)nnz * 8;
}

size_t sage_bcm_prep_bytes(int64_t n_cols, int64_t width, bool hbf16) { return geometry_b(64, n_cols, width, hbf16).prep_bytes; }

size_t sage_bcm_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, bool hbf16, int64_t nnz) {
  return sage_bcm_prep_bytes(n_cols, width, hbf16) + sage_bcm_plan_bytes(n_dst, n_cols, nnz);
}

int sage_bcm_plan(int64_t n_dst, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col, const float* w, void* plan,
                  hipStream_t st) {
  const GeoB g = geometry_b(n_dst, n_cols, 32, false);
  int32_t* chunk_ptr = static_cast<int32_t*>(plan);
  u32x2* pent = reinterpret_cast<u32x2*>(static_cast<char*>(plan) + g.cptr_bytes);
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sage_bcm_pack_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024) == hipSuccess;
  if (!ok) return fail(DH_ERR_LAUNCH, "dh_sage_window_plan: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(sage_bcm_pack_kernel<false>, dim3((unsigned)ceil_div(n_dst, 64)), dim3(1024), PACK_LDS, st, n_dst, (int)col_begin, (int)n_cols,
                     32 * g.n_chunks, 1, rowptr, col, w, (const int32_t*)nullptr, chunk_ptr, pent);
  return check_launch("dh_sage_window_plan");
}

int sage_bcm_launch(int64_t n_dst, int64_t width, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                    const float* w, const float* colscale, const void* H, int64_t ldh, bool hb, void* neigh, int64_t ldn, bool ob,
                    int64_t nnz, const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha, int64_t n_genes,
                    const void* plan, void* workspace, hipStream_t st) {
  const char* me = "dh_sage_window_mfma";
  const GeoB g = geometry_b(n_dst, n_cols, width, hb);
  uint16_t* HsP = static_cast<uint16_t*>(workspace);
  const int32_t* chunk_ptr = static_cast<const int32_t*>(plan);
  const u32x2* pent = reinterpret_cast<const u32x2*>(static_cast<const char*>(plan) + g.cptr_bytes);
  const char* Hw = static_cast<const char*>(H) + (size_t)col_begin * ldh * (hb ? 2 : 4);
  const unsigned pgrid = (unsigned)ceil_div((int64_t)g.J * 2 * g.Dp, 256);
  const int64_t step_elems = g.step_bytes / 2;
  if (hb) hipLaunchKernelGGL(sage_bcm_prep_kernel<true>, dim3(pgrid), dim3(256), 0, st, n_cols, width, g.J, g.Dp, Hw, ldh, HsP, step_elems);
  else hipLaunchKernelGGL(sage_bcm_prep_kernel<false>, dim3(pgrid), dim3(256), 0, st, n_cols, width, g.J, g.Dp, Hw, ldh, HsP, step_elems);
  const unsigned grid = (unsigned)ceil_div(n_dst, 128);
#define DH_SB(HB, OB, NTV)                                                                                                         \
  do {                                                                                                                             \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sage_bcm_kernel<HB, OB, NTV, false>),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;              \
    if (!ok) return fail(DH_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit", me);                                             \
    hipLaunchKernelGGL((sage_bcm_kernel<HB, OB, NTV, false>), dim3(grid), dim3(512), g.lds_bytes, st, n_dst, width, (int)col_begin, \
                       (int)n_cols, 32 * g.n_chunks, 1, 1, MAX_CHUNKS + 4, g.Dp, rowptr, col, w, colscale, chunk_ptr, pent, HsP, neigh, ldn, nnz,    \
                       g.step_bytes, H, ldh, src_cell_id, dst_cell_id, alpha, (int)n_genes, (float*)nullptr);                      \
  } while (0)
#define DH_SBN(HB, OB)                                                                                                             \
  do {                                                                                                                             \
    if (g.nt == 1) DH_SB(HB, OB, 1);                                                                                               \
    else if (g.nt == 2) DH_SB(HB, OB, 2);                                                                                          \
    else if (g.nt == 3) DH_SB(HB, OB, 3);                                                                                          \
    else DH_SB(HB, OB, 4);                                                                                                         \
  } while (0)
  if (hb && ob) DH_SBN(true, true);
  else if (hb) DH_SBN(true, false);
  else if (ob) DH_SBN(false, true);
  else DH_SBN(false, false);
#undef DH_SBN
#undef DH_SB
  return check_launch(me);
}


}  // namespace dh

/* ---- windows too wide for one pass (gene <- cell: 2000 destination rows, a window of every cell) --------------------------------------
 * The same loop with the K dimension cut into slices of SPLIT_COLS columns and the slices dealt to `sets` of workgroups; see
 * sage_bcm_kernel<.., SPLITK>.  Plan = [chunk pointers per (64-row group, slice)] [row bounds per (row, slice)] [packed entries]. */
namespace {

constexpr int SPLIT_COLS = 2048;  // 64 chunks (the kernel's chunk-pointer index c + c / 64 knows): 10 % density gives a row ~205 entries per slice (the pack kernel's register path holds 256)

struct GeoS {
  GeoB b;  // of ONE slice: LDS plan, Dp, nt, step_bytes
  int S, n_sets, slices_per_set, row_blocks, cps1, cptr_stride;
  int64_t J_total;
  size_t cptr_bytes, bounds_bytes, prep_bytes, partial_bytes;
};
GeoS geometry_s(int64_t n_dst, int64_t n_cols, int64_t width, bool hbf16) {
  GeoS g;
  g.b = geometry_b(n_dst, n_cols < SPLIT_COLS ? n_cols : SPLIT_COLS, width, hbf16);
  g.S = (int)dh::ceil_div(n_cols, (int64_t)SPLIT_COLS);
  g.row_blocks = (int)dh::ceil_div(n_dst, (int64_t)128);
  // one workgroup per CU (256 of them), in whole multiples of 8 sets (a set's row blocks share an XCD), no more sets than slices;
  // more sets (= more rounds of workgroups) when a set's chunk-pointer table would not fit LDS next to the images (very wide windows)
  g.cps1 = SPLIT_COLS / 32 + 1;
  const size_t lds_fixed = 128 * 16 + 256 * 4 + (size_t)2 * 2 * IMG * 2 + (size_t)3 * (hbf16 ? g.b.nt : 2 * g.b.nt) * 4096;
  int sets = 256 / g.row_blocks / 8 * 8;
  if (sets < 8) sets = 8;
  if (sets > g.S) sets = g.S;
  auto stride_of = [&](int n_sets) { return (((g.S + n_sets - 1) / n_sets) * g.cps1 + 4 + 3) / 4 * 4; };
  while (sets < g.S && lds_fixed + (size_t)2 * stride_of(sets) * 4 > 160 * 1024) sets = sets + 8 < g.S ? sets + 8 : g.S;
  g.slices_per_set = (g.S + sets - 1) / sets;
  g.n_sets = (g.S + g.slices_per_set - 1) / g.slices_per_set;
  g.J_total = (n_cols + 15) / 16;
  g.cptr_bytes = ((size_t)dh::ceil_div(n_dst, 64) * g.S * g.cps1 * 4 + 255) / 256 * 256;
  g.bounds_bytes = ((size_t)n_dst * (g.S + 1) * 4 + 255) / 256 * 256;
  g.prep_bytes = ((size_t)g.J_total * g.b.step_bytes + 255) / 256 * 256;
  g.partial_bytes = (size_t)g.n_sets * n_dst * g.b.Dp * 4;
  g.cptr_stride = (g.slices_per_set * g.cps1 + 4 + 3) / 4 * 4;  // (a multiple of 16 bytes: the A images and feature slots behind the table are read 16 bytes at a time)
  g.b.lds_bytes = lds_fixed + (size_t)2 * g.cptr_stride * 4;
  return g;
}
bool splitk_fits(int64_t n_dst, int64_t n_cols, int64_t width, int64_t nnz) {
  return n_dst > 0 && n_dst <= 65536 && n_cols > 0 && n_cols < ((int64_t)1 << 30) && width > 0 && width <= 512 && nnz >= 2 && nnz < ((int64_t)1 << 31);
}

}  // namespace

extern "C" size_t dh_sage_window_splitk_plan_bytes(int64_t n_dst, int64_t n_cols, int64_t nnz) {
  if (!splitk_fits(n_dst, n_cols, 32, nnz)) return 0;
  const GeoS g = geometry_s(n_dst, n_cols, 32, false);
  return g.cptr_bytes + g.bounds_bytes + (size_t)nnz * 8;
}

extern "C" int dh_sage_window_splitk_plan(int64_t n_dst, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                                          const float* w, int64_t nnz, void* plan, size_t plan_bytes, dh_stream_t stream) {
  const char* me = "dh_sage_window_splitk_plan";
  if (n_dst < 0 || col_begin < 0 || n_cols <= 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "%s: bad size", me);
  if (n_dst == 0) return DH_OK;
  if (!rowptr || !col || !w || !plan) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (!splitk_fits(n_dst, n_cols, 32, nnz)) return dh::fail(DH_ERR_INVALID, "%s: %lld rows / %lld columns / %lld entries not supported", me, (long long)n_dst, (long long)n_cols, (long long)nnz);
  const GeoS g = geometry_s(n_dst, n_cols, 32, false);
  const size_t need = g.cptr_bytes + g.bounds_bytes + (size_t)nnz * 8;
  if (plan_bytes < need) return dh::fail(DH_ERR_WORKSPACE, "%s: plan buffer %zu < %zu bytes", me, plan_bytes, need);
  hipStream_t st = dh::as_stream(stream);
  int32_t* chunk_ptr = static_cast<int32_t*>(plan);
  int32_t* bounds = reinterpret_cast<int32_t*>(static_cast<char*>(plan) + g.cptr_bytes);
  u32x2* pent = reinterpret_cast<u32x2*>(static_cast<char*>(plan) + g.cptr_bytes + g.bounds_bytes);
  const int64_t nb = n_dst * (g.S + 1);
  hipLaunchKernelGGL(sage_bcm_bounds_kernel, dim3((unsigned)dh::ceil_div(nb, (int64_t)256)), dim3(256), 0, st, n_dst, (int)col_begin, (int)n_cols,
                     SPLIT_COLS, g.S, rowptr, col, bounds);
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sage_bcm_pack_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024) == hipSuccess;
  if (!ok) return dh::fail(DH_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit", me);
  hipLaunchKernelGGL(sage_bcm_pack_kernel<true>, dim3((unsigned)dh::ceil_div(n_dst, (int64_t)64), (unsigned)g.S), dim3(1024), PACK_LDS, st, n_dst,
                     (int)col_begin, (int)n_cols, SPLIT_COLS, g.S, rowptr, col, w, bounds, chunk_ptr, pent);
  return dh::check_launch(me);
}

extern "C" int dh_sage_window_splitk_supported(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype, const void* H, int64_t ldh, int64_t nnz) {
  (void)H;
  (void)ldh;
  if (!splitk_fits(n_dst, n_cols, width, nnz)) return 0;
  return geometry_s(n_dst, n_cols, width, h_dtype == DH_DTYPE_BF16).b.lds_bytes <= 160 * 1024 ? 1 : 0;
}

extern "C" size_t dh_sage_window_splitk_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype) {
  if (!splitk_fits(n_dst, n_cols, width, 2)) return 0;
  const GeoS g = geometry_s(n_dst, n_cols, width, h_dtype == DH_DTYPE_BF16);
  return g.prep_bytes + g.partial_bytes;
}

extern "C" int dh_sage_window_splitk(int64_t n_dst, int64_t n_src, int64_t width, int64_t col_begin, int64_t n_cols, const int32_t* rowptr,
                                     const int32_t* col, const float* w, const float* rowscale, const void* H, int64_t ldh,
                                     int h_dtype, void* neigh, int64_t ldn, int out_dtype, int64_t nnz,
                                     const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha, int64_t n_genes,
                                     const void* plan, size_t plan_bytes, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_sage_window_splitk";
  if (n_dst < 0 || n_src < 0 || width < 0 || col_begin < 0 || n_cols < 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_dst == 0 || width == 0) return DH_OK;
  if (!rowptr || !col || !w || !H || !neigh || !plan || !workspace) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (!src_cell_id || !dst_cell_id || !alpha) return dh::fail(DH_ERR_INVALID, "%s: src_cell_id, dst_cell_id and alpha are required (the out-of-window edges are folded in)", me);
  if (ldh < width || ldn < width) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < width", me);
  if ((h_dtype != DH_DTYPE_F32 && h_dtype != DH_DTYPE_BF16) || (out_dtype != DH_DTYPE_F32 && out_dtype != DH_DTYPE_BF16)) return dh::fail(DH_ERR_INVALID, "%s: bad dtype", me);
  if (n_cols <= 0 || col_begin + n_cols > n_src) return dh::fail(DH_ERR_INVALID, "%s: window beyond the source rows", me);
  const bool hb = h_dtype == DH_DTYPE_BF16, ob = out_dtype == DH_DTYPE_BF16;
  if (!dh_sage_window_splitk_supported(n_dst, n_cols, width, h_dtype, H, ldh, nnz)) return dh::fail(DH_ERR_INVALID, "%s: shape not supported (see dh_sage_window_splitk_supported)", me);
  const GeoS g = geometry_s(n_dst, n_cols, width, hb);
  if (plan_bytes < g.cptr_bytes + g.bounds_bytes + (size_t)nnz * 8) return dh::fail(DH_ERR_WORKSPACE, "%s: plan buffer %zu too small", me, plan_bytes);
  if (workspace_bytes < g.prep_bytes + g.partial_bytes) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, workspace_bytes, g.prep_bytes + g.partial_bytes);
  hipStream_t st = dh::as_stream(stream);
  uint16_t* HsP = static_cast<uint16_t*>(workspace);
  float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + g.prep_bytes);
  const int32_t* chunk_ptr = static_cast<const int32_t*>(plan);
  const u32x2* pent = reinterpret_cast<const u32x2*>(static_cast<const char*>(plan) + g.cptr_bytes + g.bounds_bytes);
  const char* Hw = static_cast<const char*>(H) + (size_t)col_begin * ldh * (hb ? 2 : 4);
  const unsigned pgrid = (unsigned)dh::ceil_div(g.J_total * 2 * g.b.Dp, (int64_t)256);
  const int64_t step_elems = g.b.step_bytes / 2;
  if (hb) hipLaunchKernelGGL(sage_bcm_prep_kernel<true>, dim3(pgrid), dim3(256), 0, st, n_cols, width, (int)g.J_total, g.b.Dp, Hw, ldh, HsP, step_elems);
  else hipLaunchKernelGGL(sage_bcm_prep_kernel<false>, dim3(pgrid), dim3(256), 0, st, n_cols, width, (int)g.J_total, g.b.Dp, Hw, ldh, HsP, step_elems);
  const unsigned grid = (unsigned)(((g.n_sets + 7) / 8) * 8 * g.row_blocks);
#define DH_SK(HB, NTV)                                                                                                             \
  do {                                                                                                                             \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sage_bcm_kernel<HB, false, NTV, true>),               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;              \
    if (!ok) return dh::fail(DH_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit", me);                                         \
    hipLaunchKernelGGL((sage_bcm_kernel<HB, false, NTV, true>), dim3(grid), dim3(512), g.b.lds_bytes, st, n_dst, width,            \
                       (int)col_begin, (int)n_cols, SPLIT_COLS, g.S, g.slices_per_set, g.cptr_stride, g.b.Dp, rowptr, col, w, (const float*)nullptr, chunk_ptr, \
                       pent, HsP, (void*)nullptr, (int64_t)0, nnz, g.b.step_bytes, H, ldh, src_cell_id, dst_cell_id, alpha,         \
                       (int)n_genes, partial);                                                                                     \
  } while (0)
#define DH_SKN(HB)                                                                                                                 \
  do {                                                                                                                             \
    if (g.b.nt == 1) DH_SK(HB, 1);                                                                                                 \
    else if (g.b.nt == 2) DH_SK(HB, 2);                                                                                            \
    else if (g.b.nt == 3) DH_SK(HB, 3);                                                                                            \
    else DH_SK(HB, 4);                                                                                                             \
  } while (0)
  if (hb) DH_SKN(true);
  else DH_SKN(false);
#undef DH_SKN
#undef DH_SK
  if (int rc = dh::check_launch(me)) return rc;
#define DH_RED(HB, OB)                                                                                                             \
  hipLaunchKernelGGL((sage_bcm_reduce_kernel<HB, OB>), dim3((unsigned)n_dst), dim3(256), 0, st, n_dst, width, g.b.Dp, g.n_sets,     \
                     (int)col_begin, (int)n_cols, partial, rowscale, rowptr, col, w, H, ldh, src_cell_id, dst_cell_id, alpha,       \
                     (int)n_genes, neigh, ldn)
  if (hb && ob) DH_RED(true, true);
  else if (hb) DH_RED(true, false);
  else if (ob) DH_RED(false, true);
  else DH_RED(false, false);
#undef DH_RED
  return dh::check_launch(me);
}
