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
// in the 32-lanes-per-row float4 configuration (128-column passes, two rows per wavefront) the forward pass stores, per
// row and per 128-column slice, the bitmap "column c of the slice is > 0" (4 x uint32 = 128 bits, bit c % 32 of word
// c / 32); the backward pass gathers dY rows and zeroes the elements whose bit is clear, so G = dY * [Y > 0] is never
// written to or read from HBM.
struct ReluMask {
  uint32_t* out;       // forward: written when non-null   [n_rows][slices][4]
  const uint32_t* in;  // backward: applied to the gathered rows of Z when non-null [n_cols][slices][4]
  int slices;          // width / 128 of the whole layer
  int slice0;          // 128-column slice of this launch (column-sliced passes)
  int xcd_blocks;      // > 0: blocks per XCD of the XCD-contiguous row mapping (see spmm_slice128_kernel); 0: block b = rows 8 b ..
};

// G lanes per row, VEC floats per lane per slice, NACC slices per lane (slices G*VEC apart), BATCH neighbour rows requested before the
// first is used.  BATCH = 4 is the setting of the HBM-bound graphs; a block of a mini-batch (129 rows of 201 entries) is a chain of
// dependent round trips instead — 50 batches of 4 = 21 us per launch — and runs with BATCH = 16.  The accumulation order, hence every
// bit of the result, is the same for any BATCH.
template <int G, int VEC, int NACC, bool SAGE, int BATCH>
__global__ __launch_bounds__(256) void spmm_csr_kernel(
    int64_t n_rows, int64_t width, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ rowscale, const float* __restrict__ colscale,
    const float* __restrict__ Z, int64_t ldz, float* __restrict__ Y, int64_t ldy,
    const float* __restrict__ bias, int act, int reduce, SageScale sage, const int32_t* __restrict__ row_ids) {
  using V = typename VecT<VEC>::type;
  constexpr int ROWS_PER_BLOCK = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t slot = (int64_t)blockIdx.x * ROWS_PER_BLOCK + threadIdx.x / G;
  if (slot >= n_rows) return;  // whole groups exit together
  // row_ids: the rows this launch computes (interior / boundary subsets of a shard, sharding.py); null = rows [0, n_rows)
  const int64_t row = row_ids ? (int64_t)row_ids[slot] : slot;

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
    // BATCH neighbour rows in flight per lane per slice
    for (; k + BATCH <= cnt; k += BATCH) {
      V z[BATCH][NACC];
      float wk[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int ck = bcast_i<G>(c, k + u);
        wk[u] = bcast_f<G>(w, k + u);
        const float* zr = Z + (int64_t)ck * ldz + c0;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
          z[u][a] = live[a] ? *reinterpret_cast<const V*>(zr + a * G * VEC) : V(0.f);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
#pragma unroll
        for (int a = 0; a < NACC; ++a) fma_vec<VEC>(acc[a], wk[u], z[u][a]);
    }
    if constexpr (BATCH > 4) {  // what is left of the 64-entry chunk: batches of 4, then singles
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
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int a = 0; a < NACC; ++a) fma_vec<VEC>(acc[a], wk[u], z[u][a]);
      }
    }
    for (; k < cnt; ++k) {
      const int ck = bcast_i<G>(c, k);
      const float wk = bcast_f<G>(w, k);
      const float* zr = Z + (int64_t)ck * ldz + c0;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        if (!live[a]) continue;
        V zv = *reinterpret_cast<const V*>(zr + a * G * VEC);
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
    __builtin_nontemporal_store(acc[a], reinterpret_cast<V*>(yr + a * G * VEC));
  }
}

// One 128-column slice of a wide layer: 32 lanes x float4 per row, two rows per wavefront, every lane live.  This is the
// kernel of the headline layer (four passes per SpMM).  MIN: the gathered rows are dY and the ReLU sign mask of the same
// row is applied on the fly (backward); MOUT: the sign mask of the output is recorded (forward).  IDX32: n_cols * ldz fits
// 32 bits, so a neighbour's row offset is one 32-bit multiply.  All loads of a batch of 4 neighbours (4 x 16 B of Z, and with
// MIN 4 x 16 B of mask) are issued before the first use: the version this replaces applied the mask per neighbour behind a
// run-time branch and drained vmcnt after every mask load (one neighbour in flight; backward 5.4 ms vs 4.6 ms forward).
template <bool MIN, bool MOUT, bool IDX32>
__global__ __launch_bounds__(256) void spmm_slice128_kernel(
    int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ rowscale, const float* __restrict__ colscale, const float* __restrict__ Z, int64_t ldz,
    float* __restrict__ Y, int64_t ldy, const float* __restrict__ bias, int act, int reduce,
    const int32_t* __restrict__ row_ids, ReluMask mask) {
  const int g = threadIdx.x & 31;
  // The dispatcher places block b on XCD b % 8.  With xcd_blocks > 0 XCD x walks its OWN contiguous eighth of the rows
  // (blocks x * xcd_blocks ...): on a locality-ordered graph (CSRGraph.permute) the rows gathered by neighbouring
  // destination rows overlap, and this way they overlap inside one XCD's private L2 instead of being spread over all eight.
  int64_t blk = blockIdx.x;
  if (mask.xcd_blocks > 0) blk = (int64_t)(blockIdx.x & 7) * mask.xcd_blocks + (blockIdx.x >> 3);
  const int64_t slot = blk * 8 + (threadIdx.x >> 5);
  if (slot >= n_rows) return;  // whole 32-lane groups exit together
  const int64_t row = row_ids ? (int64_t)row_ids[slot] : slot;
  const float* zc = Z + g * 4;
  const uint32_t* min_base = MIN ? mask.in + (int64_t)mask.slice0 * 4 : nullptr;
  auto zrow = [&](int ck) -> const f32x4* {
    if constexpr (IDX32) return reinterpret_cast<const f32x4*>(zc + (uint32_t)ck * (uint32_t)ldz);
    else return reinterpret_cast<const f32x4*>(zc + (int64_t)ck * ldz);
  };
  // the mask of a (row, slice) is a plain 128-bit bitmap (bit c = column c of the slice): this lane's four columns
  // 4 g .. 4 g + 3 are bits 4 (g & 7) .. + 3 of word g >> 3 — one 4-byte load per neighbour (a 16-byte load per lane cost
  // the texture path as much as the 16 bytes of data it guards: 5.35 vs 4.49 ms unmasked)
  const uint32_t* min_lane = MIN ? min_base + (g >> 3) : nullptr;
  const uint32_t bit0 = 4u * (uint32_t)(g & 7);
  auto mrow = [&](int ck) -> uint32_t { return min_lane[(int64_t)ck * (mask.slices * 4)]; };
  auto masked = [&](f32x4 z, uint32_t m) -> f32x4 {  // v_bfe_i32 (bit -> 0 / ~0) + v_and per element
#pragma unroll
    for (int i = 0; i < 4; ++i)
      z[i] = __uint_as_float(__float_as_uint(z[i]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, bit0 + i, 1u));
    return z;
  };

  f32x4 acc = f32x4(0.f);
  const int s = rowptr[row], t = rowptr[row + 1];
  for (int base = s; base < t; base += 32) {
    const int e = base + g;
    int c = 0;
    float w = 0.f;
    if (e < t) {
      c = col[e];
      w = val ? val[e] : 1.f;
      if (colscale) w *= colscale[c];
    }
    const int cnt = min(32, t - base);
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
      int ck[4];
      float wk[4];
      f32x4 z[4];
      uint32_t m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ck[u] = __shfl(c, k + u, 32);
        wk[u] = __shfl(w, k + u, 32);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) z[u] = *zrow(ck[u]);
      if constexpr (MIN) {
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = mrow(ck[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) z[u] = masked(z[u], m[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) fma_vec<4>(acc, wk[u], z[u]);
    }
    const int rem = cnt - k;  // 0..3 neighbours left: their loads go out together as well
    if (rem > 0) {
      int ck[3];
      float wk[3];
      f32x4 z[3];
      uint32_t m[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        ck[u] = __shfl(c, k + u, 32);  // lanes beyond cnt hold c = 0, w = 0 (never used below)
        wk[u] = __shfl(w, k + u, 32);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (u < rem) {
          z[u] = *zrow(ck[u]);
          if constexpr (MIN) m[u] = mrow(ck[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (u < rem) {
          if constexpr (MIN) z[u] = masked(z[u], m[u]);
          fma_vec<4>(acc, wk[u], z[u]);
        }
      }
    }
  }

  float scale = rowscale ? rowscale[row] : 1.f;
  if (reduce == DH_REDUCE_MEAN) scale = (t > s) ? scale / (float)(t - s) : 0.f;
  epilogue<4>(acc, scale, bias, (int64_t)g * 4, act);
  if constexpr (MOUT) {
    uint32_t nib = (acc[0] > 0.f ? 1u : 0u) | (acc[1] > 0.f ? 2u : 0u) | (acc[2] > 0.f ? 4u : 0u) | (acc[3] > 0.f ? 8u : 0u);
    nib <<= 4 * (g & 7);
    nib |= __shfl_xor(nib, 1, 8);  // OR over the 8 lanes that share a word
    nib |= __shfl_xor(nib, 2, 8);
    nib |= __shfl_xor(nib, 4, 8);
    if ((g & 7) == 0) mask.out[(row * mask.slices + mask.slice0) * 4 + (g >> 3)] = nib;
  }
  __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(Y + row * ldy + g * 4));
}

// Resident form of spmm_slice128_kernel for the co-scheduled layer (autograd.py: the GEMM of column panel c + 1 next to the
// aggregation of panel c).  A GCN layer's GEMM is bound by the matrix cores, its aggregation by HBM; they use disjoint units
// of a CU, but two ordinary launches do not overlap: the dispatcher hands every register a retiring GEMM workgroup frees to the
// small aggregation workgroups queued behind it, the next GEMM workgroup (176 registers on each of the 4 SIMDs + 72 KB of LDS at
// once) never finds room, and the CUs drift from one kernel to the other — the two-stream layer of round 3 measured exactly
// serial time (profiles/r03b_pipeline_probe.json).  This form takes a FIXED footprint instead: `gridDim.x` workgroups of 512
// threads (two waves per SIMD, <= 80 registers each = the 160 registers per SIMD the 128 x 128 GEMM configuration leaves free;
// a second one does not fit beside two GEMM workgroups, so the dispatcher can only place one per CU), each walking the 16-row
// blocks b, b + gridDim.x, ... of the slice.  Nothing else is ever dispatched into a freed GEMM slot but the next GEMM workgroup.
// With two waves per SIMD instead of eight, latency is hidden by depth instead of occupancy: the (col, val) chunk of the next
// row and the row pointers of the one after are fetched while the current row's gathers are in flight, and a row's
// neighbours are requested 8 at a time.  Accumulation order = CSR order: bit-identical to spmm_slice128_kernel.
template <bool MIN, bool MOUT, bool IDX32, int NB, int THREADS>
__global__ __launch_bounds__(THREADS) void spmm_slice128_resident_kernel(
    int n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ Z, int64_t ldz, float* __restrict__ Y, int64_t ldy, const float* __restrict__ bias, int act,
    const int32_t* __restrict__ row_ids, ReluMask mask) {
  // NB = neighbour rows requested before the first is used; THREADS / 32 rows per block
  constexpr int RPB = THREADS / 32;
  const int g = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;
  const float* zc = Z + g * 4;
  const uint32_t* min_lane = MIN ? mask.in + (int64_t)mask.slice0 * 4 + (g >> 3) : nullptr;
  const uint32_t bit0 = 4u * (uint32_t)(g & 7);
  const uint32_t mstride = MIN ? (uint32_t)mask.slices * 4u : 0u;
  auto zrow = [&](int ck) -> const f32x4* {
    if constexpr (IDX32) return reinterpret_cast<const f32x4*>(zc + (uint32_t)ck * (uint32_t)ldz);
    else return reinterpret_cast<const f32x4*>(zc + (int64_t)ck * ldz);
  };
  auto mrow = [&](int ck) -> uint32_t {
    if constexpr (IDX32) return min_lane[(uint32_t)ck * mstride];
    else return min_lane[(int64_t)ck * mstride];
  };
  auto masked = [&](f32x4 z, uint32_t m) -> f32x4 {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      z[i] = __uint_as_float(__float_as_uint(z[i]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, bit0 + i, 1u));
    return z;
  };
  // block walk: with xcd_blocks > 0 the workgroups of XCD x (= blockIdx.x % 8, observed placement) walk XCD x's contiguous
  // eighth of the blocks, as the one-shot kernel does; otherwise block b, b + gridDim.x, ...
  const int n_blocks = (n_rows + RPB - 1) / RPB;
  int blk, blk_end, blk_step;
  if (mask.xcd_blocks > 0 && (gridDim.x & 7) == 0) {
    const int per = (n_blocks + 7) / 8;
    blk = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    blk_end = min((int)((blockIdx.x & 7) + 1) * per, n_blocks);
    blk_step = (int)(gridDim.x >> 3);
  } else {
    blk = (int)blockIdx.x;
    blk_end = n_blocks;
    blk_step = (int)gridDim.x;
  }
  auto row_of = [&](int b) -> int {  // -1: no row (past the end); b may overflow past blk_end only by 2 steps (host keeps it in range)
    const int slot = b * RPB + sub;
    if (b >= blk_end || slot >= n_rows) return -1;
    return row_ids ? row_ids[slot] : slot;
  };
  auto chunk = [&](int base, int t, int& c, float& w) {
    const int e = base + g;
    c = 0;
    w = 0.f;
    if (e < t) {
      c = col[e];
      w = val ? val[e] : 1.f;
    }
  };
  // pipeline registers: current row (row, s, t, c, w), next row (row1, s1, t1), the row after (fetched inside the loop)
  int row = row_of(blk), row1 = row_of(blk + blk_step);
  int s = 0, t = 0, s1 = 0, t1 = 0, c, c1;
  float w, w1;
  if (row >= 0) { s = rowptr[row]; t = rowptr[row + 1]; }
  if (row1 >= 0) { s1 = rowptr[row1]; t1 = rowptr[row1 + 1]; }
  chunk(s, t, c, w);
  for (; blk < blk_end; blk += blk_step) {
    // issue the next row's first chunk and the row pointers of the one after before this row's gathers
    chunk(s1, t1, c1, w1);
    const int row2 = row_of(blk + 2 * blk_step);
    int s2 = 0, t2 = 0;
    if (row2 >= 0) { s2 = rowptr[row2]; t2 = rowptr[row2 + 1]; }
    if (row >= 0) {
      f32x4 acc = f32x4(0.f);
      for (int base = s; base < t; base += 32) {
        if (base != s) chunk(base, t, c, w);
        const int cnt = min(32, t - base);
        for (int k = 0; k < cnt; k += NB) {
          const int live = min(NB, cnt - k);
          f32x4 z[NB];
          uint32_t m[NB];
          // lanes beyond cnt hold c = 0, w = 0: row 0 of Z is read and multiplied by w = 0 ... which is NOT a no-op for inf / nan
          // rows, so the loads and the fmas of the dead slots are skipped, not neutralised
#pragma unroll
          for (int u = 0; u < NB; ++u)
            if (u < live) {
              const int ck = __shfl(c, (k + u) & 31, 32);
              z[u] = *zrow(ck);
              if constexpr (MIN) m[u] = mrow(ck);
            }
#pragma unroll
          for (int u = 0; u < NB; ++u)
            if (u < live) {
              if constexpr (MIN) z[u] = masked(z[u], m[u]);
              fma_vec<4>(acc, __shfl(w, (k + u) & 31, 32), z[u]);
            }
        }
      }
      epilogue<4>(acc, 1.f, bias, (int64_t)g * 4, act);
      if constexpr (MOUT) {
        uint32_t nib = (acc[0] > 0.f ? 1u : 0u) | (acc[1] > 0.f ? 2u : 0u) | (acc[2] > 0.f ? 4u : 0u) | (acc[3] > 0.f ? 8u : 0u);
        nib <<= 4 * (g & 7);
        nib |= __shfl_xor(nib, 1, 8);
        nib |= __shfl_xor(nib, 2, 8);
        nib |= __shfl_xor(nib, 4, 8);
        if ((g & 7) == 0) mask.out[((int64_t)row * mask.slices + mask.slice0) * 4 + (g >> 3)] = nib;
      }
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(Y + (int64_t)row * ldy + g * 4));
    }
    row = row1; s = s1; t = t1; c = c1; w = w1;
    row1 = row2; s1 = s2; t1 = t2;
  }
}

// launches the slice kernel over every 128-column slice of [0, width)
template <bool MIN, bool MOUT>
void launch_slices(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col, const float* val,
                   const float* rowscale, const float* colscale, const float* Z, int64_t ldz, float* Y, int64_t ldy,
                   const float* bias, int act, int reduce, const int32_t* row_ids, const uint32_t* in_mask, uint32_t* out_mask,
                   hipStream_t st, int64_t slice_begin = 0, int64_t slice_end = -1) {
  const bool idx32 = n_cols >= 0 && (double)n_cols * (double)ldz < 4294967296.0;
  const int64_t c_end = slice_end < 0 ? width : slice_end * 128;
  // XCD-contiguous rows (see the kernel): measured at 1M rows on knn-k15, unordered 7.81 -> 7.39 ms, RCM-ordered 5.80 -> 5.73 ms,
  // rand-k15 unchanged (4.55 ms) — profiles/r03c_locality_map{0,1}.json
  const int64_t blocks = dh::ceil_div(n_rows, 8);
  const int xcd_blocks = blocks >= 64 ? (int)dh::ceil_div(blocks, 8) : 0;
  for (int64_t c = slice_begin * 128; c < c_end; c += 128) {
    const ReluMask mask{out_mask, in_mask, (int)(width / 128), (int)(c / 128), xcd_blocks};
    dim3 grid((unsigned)(xcd_blocks ? (int64_t)xcd_blocks * 8 : blocks), 1);
    if (idx32)
      hipLaunchKernelGGL((spmm_slice128_kernel<MIN, MOUT, true>), grid, dim3(256), 0, st, n_rows, rowptr, col, val, rowscale, colscale,
                         Z + c, ldz, Y + c, ldy, bias ? bias + c : nullptr, act, reduce, row_ids, mask);
    else
      hipLaunchKernelGGL((spmm_slice128_kernel<MIN, MOUT, false>), grid, dim3(256), 0, st, n_rows, rowptr, col, val, rowscale, colscale,
                         Z + c, ldz, Y + c, ldy, bias ? bias + c : nullptr, act, reduce, row_ids, mask);
  }
}

// the resident form over the slices [slice_begin, slice_end): `workgroups` resident blocks per slice.  shape 0: 512 threads, two
// waves per SIMD at <= 80 registers, 8 neighbours requested at a time; shape 1: 256 threads, one wave per SIMD at <= 160
// registers, 16 at a time (a whole k = 15 row in one round trip).  Either fits once beside two 128 x 128 GEMM workgroups.
template <bool MIN, bool MOUT>
void launch_slices_resident(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col, const float* val,
                            const float* Z, int64_t ldz, float* Y, int64_t ldy, const float* bias, int act, const int32_t* row_ids,
                            const uint32_t* in_mask, uint32_t* out_mask, hipStream_t st, int64_t slice_begin, int64_t slice_end,
                            int workgroups, int shape) {
  const int64_t max_stride = MIN ? ((ldz > width / 32) ? ldz : width / 32) : ldz;
  const bool idx32 = n_cols >= 0 && (double)n_cols * (double)max_stride < 4294967296.0;
  const int rpb = shape == 1 ? 8 : 16;
  const int64_t blocks = dh::ceil_div(n_rows, rpb);
  const int grid = (int)(blocks < workgroups ? blocks : workgroups);
  const int xcd_blocks = (blocks >= 64 && grid % 8 == 0) ? (int)dh::ceil_div(blocks, 8) : 0;
  for (int64_t c = slice_begin * 128; c < slice_end * 128; c += 128) {
    const ReluMask mask{out_mask, in_mask, (int)(width / 128), (int)(c / 128), xcd_blocks};
#define DH_RES(I32, NB, TH)                                                                                                          \
  hipLaunchKernelGGL((spmm_slice128_resident_kernel<MIN, MOUT, I32, NB, TH>), dim3(grid), dim3(TH), 0, st, (int)n_rows, rowptr, col, val, \
                     Z + c, ldz, Y + c, ldy, bias ? bias + c : nullptr, act, row_ids, mask)
    if (shape == 1) {
      if (idx32) DH_RES(true, 16, 256);
      else DH_RES(false, 16, 256);
    } else {
      if (idx32) DH_RES(true, (MIN ? 6 : 8), 512);
      else DH_RES(false, (MIN ? 6 : 8), 512);
    }
#undef DH_RES
  }
}

template <int VEC, bool SAGE>
int launch_vec(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col,
               const float* val, const float* rowscale, const float* colscale, const float* Z,
               int64_t ldz, float* Y, int64_t ldy, const float* bias, int act, int reduce,
               SageScale sage, const int32_t* row_ids, hipStream_t st) {
  const int64_t vecs = dh::ceil_div(width, VEC);
  const bool few_rows = n_rows <= 2048;  // a mini-batch block: latency-bound, deeper request batches (same arithmetic)
#define DH_SPMM_LAUNCH(G, NACC)                                                                  \
  do {                                                                                           \
    dim3 grid((unsigned)dh::ceil_div(n_rows, 256 / G),                                           \
              (unsigned)dh::ceil_div(vecs, (int64_t)G * NACC));                                  \
    if (few_rows && G >= 16)                                                                     \
      hipLaunchKernelGGL((spmm_csr_kernel<G, VEC, NACC, SAGE, 16>), grid, dim3(256), 0, st, n_rows, \
                         width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act,   \
                         reduce, sage, row_ids);                                                   \
    else                                                                                          \
      hipLaunchKernelGGL((spmm_csr_kernel<G, VEC, NACC, SAGE, 4>), grid, dim3(256), 0, st, n_rows,  \
                         width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act,   \
                         reduce, sage, row_ids);                                                   \
  } while (0)
  if (VEC == 4 && !SAGE && vecs >= 64 && vecs % 32 == 0) {
    // Wide rows: one pass per 128-column slice instead of one launch over the whole width.  Every pass gathers from a
    // working set of n_cols x 512 bytes instead of n_cols x width x 4, which the L2 / MALL hold a larger share of.
    // Measured at 1M rows, k = 15 (scripts/spmm_width_probe.py): one 512-wide launch 5.44 ms, two 256-wide passes
    // 4.90 ms, four 128-wide passes (32 lanes per row, two rows per wavefront) 4.54 ms.
    if constexpr (VEC == 4 && !SAGE)
      launch_slices<false, false>(n_rows, n_cols, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, row_ids,
                                  nullptr, nullptr, st);
  } else if (vecs > 64) DH_SPMM_LAUNCH(64, 2);
  else if (vecs > 32) DH_SPMM_LAUNCH(64, 1);
  else if (vecs > 16) DH_SPMM_LAUNCH(32, 1);
  else if (vecs > 8) DH_SPMM_LAUNCH(16, 1);
  else DH_SPMM_LAUNCH(8, 1);
#undef DH_SPMM_LAUNCH
  return dh::check_launch(SAGE ? "dh_sage_aggregate_f32" : "dh_spmm_csr_f32");
}

template <bool SAGE>
int dispatch(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr, const int32_t* col, const float* val,
             const float* rowscale, const float* colscale, const float* Z, int64_t ldz, float* Y,
             int64_t ldy, const float* bias, int act, int reduce, SageScale sage, const int32_t* row_ids, hipStream_t st) {
  const bool a16 = dh::aligned16(Z) && dh::aligned16(Y) && (!bias || dh::aligned16(bias));
  const bool a8 = ((uintptr_t)Z % 8 == 0) && ((uintptr_t)Y % 8 == 0) && (!bias || (uintptr_t)bias % 8 == 0);
  if (a16 && width % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0)
    return launch_vec<4, SAGE>(n_rows, n_cols, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage, row_ids, st);
  if (a8 && width % 2 == 0 && ldz % 2 == 0 && ldy % 2 == 0)
    return launch_vec<2, SAGE>(n_rows, n_cols, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage, row_ids, st);
  return launch_vec<1, SAGE>(n_rows, n_cols, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, sage, row_ids, st);
}

}  // namespace

// Rows variant: computes the n_list rows listed in row_ids (null: rows [0, n_list)) of Y = act(rowscale * reduce(...) + bias).
// rowptr / rowscale / Y are indexed by the row id itself.  Used by the sharded layer to run the rows that need no remote
// operand while the halo exchange is still in flight (sharding.py).
extern "C" int dh_spmm_csr_rows_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width,
                                    const int32_t* rowptr, const int32_t* col, const float* val,
                                    const float* rowscale, const float* colscale, const float* Z,
                                    int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                                    int reduce, dh_stream_t stream) {
  if (n_list < 0 || n_cols < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: negative size");
  if (n_list == 0 || width == 0) return DH_OK;
  if (!rowptr || !Z || !Y) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: null rowptr/Z/Y");
  if (ldz < width || ldy < width) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: leading dimension < width");
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: bad act %d", act);
  if (reduce != DH_REDUCE_SUM && reduce != DH_REDUCE_MEAN) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: bad reduce %d", reduce);
  if (n_list >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_f32: n_rows >= 2^31");
  return dispatch<false>(n_list, n_cols, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce,
                         SageScale{nullptr, nullptr, nullptr, 0}, row_ids, dh::as_stream(stream));
}

extern "C" int dh_spmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t width,
                               const int32_t* rowptr, const int32_t* col, const float* val,
                               const float* rowscale, const float* colscale, const float* Z,
                               int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                               int reduce, dh_stream_t stream) {
  return dh_spmm_csr_rows_f32(n_rows, nullptr, n_cols, width, rowptr, col, val, rowscale, colscale, Z, ldz, Y, ldy, bias, act, reduce, stream);
}

extern "C" size_t dh_relu_mask_bytes(int64_t n_rows, int64_t width) {
  if (n_rows <= 0 || width <= 0 || width % 128 != 0) return 0;
  return (size_t)n_rows * (size_t)(width / 128) * 4 * sizeof(uint32_t);
}

// SpMM with the ReLU of a GCN layer fused on both sides (see ReluMask): forward = dh_spmm_csr_f32(act = relu)
// that additionally records the sign mask of its output; backward = SpMM whose gathered operand is dY with the
// recorded mask applied on the fly.  Only for the float4 configuration: width % 128 == 0 and 16-byte aligned operands
// (the host falls back to dh_relu_backward_f32 + dh_spmm_csr_f32 otherwise).  One pass per 128-column slice (32 lanes
// per row, two rows per wavefront): the gathered working set of a pass is n_cols x 512 bytes, of which L2 / MALL hold a
// larger share than of 256- or 512-column rows (measured 4.51 vs 4.90 vs 5.44 ms at 1M rows, k = 15, scripts/spmm_width_probe.py).
// in_mask rows are indexed by the gathered COLUMN id: a caller whose operand has rows without a mask (halo rows that
// arrive already masked) sets their words to all-ones.
extern "C" int dh_spmm_csr_relu_rows_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width, const int32_t* rowptr,
                                         const int32_t* col, const float* val, const float* Z, int64_t ldz, float* Y,
                                         int64_t ldy, const float* bias, int act, void* out_mask, const void* in_mask,
                                         dh_stream_t stream) {
  return dh_spmm_csr_relu_slices_f32(n_list, row_ids, n_cols, width, 0, width / 128, rowptr, col, val, Z, ldz, Y, ldy, bias, act, out_mask,
                                     in_mask, stream);
}

// Column slices [slice_begin, slice_end) (units of 128 columns) of the fused-ReLU SpMM over a width-wide layer: Z, Y, bias and
// the masks are those of the WHOLE layer.  One call per slice range lets a caller aggregate slice c on one stream while the
// GEMM that produces slice c + 1 runs on another (autograd.py: the software-pipelined layer).
extern "C" int dh_spmm_csr_relu_slices_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width, int64_t slice_begin,
                                           int64_t slice_end, const int32_t* rowptr, const int32_t* col, const float* val, const float* Z,
                                           int64_t ldz, float* Y, int64_t ldy, const float* bias, int act, void* out_mask,
                                           const void* in_mask, dh_stream_t stream) {
  return dh_spmm_csr_relu_slices_resident_f32(n_list, row_ids, n_cols, width, slice_begin, slice_end, rowptr, col, val, Z, ldz, Y, ldy, bias, act,
                                              out_mask, in_mask, 0, stream);
}

// resident_workgroups > 0: every slice runs as that many resident 512-thread workgroups walking the rows
// (spmm_slice128_resident_kernel: the fixed-footprint form that co-schedules with the 128 x 128 GEMM); 0: the one-shot grid.
// The results are bit-identical either way.
extern "C" int dh_spmm_csr_relu_slices_resident_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width, int64_t slice_begin,
                                                    int64_t slice_end, const int32_t* rowptr, const int32_t* col, const float* val,
                                                    const float* Z, int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                                                    void* out_mask, const void* in_mask, int resident_workgroups, dh_stream_t stream) {
  if (resident_workgroups < 0) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: negative resident_workgroups");
  if (slice_begin < 0 || slice_end < slice_begin || slice_end * 128 > width)
    return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: slice range [%lld, %lld) outside width %lld", (long long)slice_begin,
                    (long long)slice_end, (long long)width);
  if (n_list < 0 || n_cols < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: negative size");
  if (n_list == 0 || width == 0) return DH_OK;
  if (!rowptr || !Z || !Y) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: null rowptr/Z/Y");
  if (width % 128 != 0 || ldz % 4 != 0 || ldy % 4 != 0 || !dh::aligned16(Z) || !dh::aligned16(Y) || (bias && !dh::aligned16(bias)) ||
      (in_mask && !dh::aligned16(in_mask)))
    return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: needs width %% 128 == 0 and 16-byte aligned rows / masks");
  if (slice_end == slice_begin) return DH_OK;
  if (ldz < width || ldy < width) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: leading dimension < width");
  if (act != DH_ACT_NONE && act != DH_ACT_RELU) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: bad act %d", act);
  if (n_list >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "dh_spmm_csr_relu_f32: n_rows >= 2^31");
  hipStream_t st = dh::as_stream(stream);
  const uint32_t* mi = static_cast<const uint32_t*>(in_mask);
  uint32_t* mo = static_cast<uint32_t*>(out_mask);
#define DH_SLICES(MIN, MOUT)                                                                                                        \
  do {                                                                                                                              \
    if (resident_workgroups > 0)                                                                                                    \
      launch_slices_resident<MIN, MOUT>(n_list, n_cols, width, rowptr, col, val, Z, ldz, Y, ldy, bias, act, row_ids, mi, mo, st,     \
                                        slice_begin, slice_end, resident_workgroups & 0xffff, resident_workgroups >> 16);           \
    else                                                                                                                            \
      launch_slices<MIN, MOUT>(n_list, n_cols, width, rowptr, col, val, nullptr, nullptr, Z, ldz, Y, ldy, bias, act, DH_REDUCE_SUM,  \
                               row_ids, mi, mo, st, slice_begin, slice_end);                                                        \
  } while (0)
  if (mi && mo) DH_SLICES(true, true);
  else if (mi) DH_SLICES(true, false);
  else if (mo) DH_SLICES(false, true);
  else DH_SLICES(false, false);
#undef DH_SLICES
  return dh::check_launch("dh_spmm_csr_relu_f32");
}

extern "C" int dh_spmm_csr_relu_f32(int64_t n_rows, int64_t n_cols, int64_t width, const int32_t* rowptr,
                                    const int32_t* col, const float* val, const float* Z, int64_t ldz, float* Y,
                                    int64_t ldy, const float* bias, int act, void* out_mask, const void* in_mask,
                                    dh_stream_t stream) {
  return dh_spmm_csr_relu_rows_f32(n_rows, nullptr, n_cols, width, rowptr, col, val, Z, ldz, Y, ldy, bias, act, out_mask, in_mask, stream);
}

// out[i,:] = X[idx[i],:] (* [mask bit] if a ReLU sign mask of X's rows is given): packs the rows a peer asked for into a
// contiguous send buffer of the halo exchange; with a mask the receiver gets G = dY * [Y > 0] rows without G ever existing.
namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(int64_t n, int64_t width, const int32_t* __restrict__ idx, const float* __restrict__ X,
                                                          int64_t ldx, const uint32_t* __restrict__ mask, int slices, float* __restrict__ out,
                                                          int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t r = idx[i];
  const float* x = X + r * ldx;
  float* o = out + i * ldo;
  for (int64_t c = (int64_t)lane * 4; c < width; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
    if (mask) {  // columns c .. c + 3 of slice c / 128: bits (c % 32) .. + 3 of word (c % 128) / 32
      const uint32_t m = mask[(r * slices + c / 128) * 4 + (c % 128) / 32] >> (c % 32);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!((m >> j) & 1u)) v[j] = 0.f;
    }
    *reinterpret_cast<f32x4*>(o + c) = v;
  }
}
}  // namespace

// G = X * [mask bit]: the streaming form of the ReLU backward when the sign mask of the layer output (not the output
// itself) was kept: one 16-byte load + one mask word per 8 lanes, one non-temporal 16-byte store.
namespace {
__global__ __launch_bounds__(256) void relu_mask_apply_kernel(int64_t n_rows, int vec_per_row, const float* __restrict__ X, int64_t ldx,
                                                              const uint32_t* __restrict__ mask, float* __restrict__ out, int64_t ldo) {
  const int64_t total = n_rows * vec_per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / vec_per_row;
    const int c = (int)(i - r * vec_per_row) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
    const uint32_t m = mask[r * (vec_per_row / 8) + (c >> 5)] >> (c & 31);  // the row's bitmap is width / 32 words
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(__float_as_uint(v[j]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, j, 1u));
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + r * ldo + c));
  }
}
}  // namespace

extern "C" int dh_relu_mask_apply_f32(int64_t n_rows, int64_t width, const float* X, int64_t ldx, const void* relu_mask, float* out,
                                      int64_t ldo, dh_stream_t stream) {
  if (n_rows < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_relu_mask_apply_f32: negative size");
  if (n_rows == 0 || width == 0) return DH_OK;
  if (!X || !out || !relu_mask) return dh::fail(DH_ERR_INVALID, "dh_relu_mask_apply_f32: null pointer");
  if (width % 128 || ldx % 4 || ldo % 4 || ldx < width || ldo < width || !dh::aligned16(X) || !dh::aligned16(out))
    return dh::fail(DH_ERR_INVALID, "dh_relu_mask_apply_f32: needs width %% 128 == 0 and 16-byte aligned rows");
  const int64_t total = n_rows * (width / 4);
  const unsigned grid = (unsigned)(dh::ceil_div(total, 256) < 16384 ? dh::ceil_div(total, 256) : 16384);
  hipLaunchKernelGGL(relu_mask_apply_kernel, dim3(grid), dim3(256), 0, dh::as_stream(stream), n_rows, (int)(width / 4), X, ldx,
                     static_cast<const uint32_t*>(relu_mask), out, ldo);
  return dh::check_launch("dh_relu_mask_apply_f32");
}

extern "C" int dh_gather_rows_f32(int64_t n, int64_t width, const int32_t* idx, const float* X, int64_t ldx, const void* relu_mask,
                                  float* out, int64_t ldo, dh_stream_t stream) {
  if (n < 0 || width < 0) return dh::fail(DH_ERR_INVALID, "dh_gather_rows_f32: negative size");
  if (n == 0 || width == 0) return DH_OK;
  if (!idx || !X || !out) return dh::fail(DH_ERR_INVALID, "dh_gather_rows_f32: null pointer");
  if (width % 4 || ldx % 4 || ldo % 4 || ldx < width || ldo < width || !dh::aligned16(X) || !dh::aligned16(out))
    return dh::fail(DH_ERR_INVALID, "dh_gather_rows_f32: needs width %% 4 == 0 and 16-byte aligned rows");
  if (relu_mask && width % 128) return dh::fail(DH_ERR_INVALID, "dh_gather_rows_f32: a ReLU mask needs width %% 128 == 0");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)dh::ceil_div(n, 4)), dim3(256), 0, dh::as_stream(stream), n, width, idx, X, ldx,
                     static_cast<const uint32_t*>(relu_mask), (int)(width / 128), out, ldo);
  return dh::check_launch("dh_gather_rows_f32");
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
  return dispatch<true>(n_dst, n_src, width, rowptr, col, w, nullptr, nullptr, H, ldh, neigh, ldn, nullptr, DH_ACT_NONE,
                        DH_REDUCE_MEAN, SageScale{src_cell_id, dst_cell_id, alpha, (int)n_genes}, nullptr, dh::as_stream(stream));
}
