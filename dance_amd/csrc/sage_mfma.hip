// AdaptiveSAGE cell <- gene aggregation on the matrix cores WITHOUT a dense adjacency in HBM (SURVEY.md §8a A3, configs
// C3 / C4; dance/models/nn/gnn.py:62-90).
//
//   neigh[v,:] += 1/deg(v) * sum over the in-edges e = (u -> v) with u inside the window [col_begin, col_begin + n_cols)
//                            of  (w_e * colscale[u - col_begin]) * H[u,:]
//
// (the window = the gene rows of H, colscale = alpha[cell_id of the gene]; every other in-edge — the self loop — is
// dh_sage_tail's, which initialises neigh).  The cell-gene graph is 10 % dense: 2e8 edges point at 2000 gene rows, so the
// gather kernels are bound by whatever delivers 320 GB of gene-row reads (L2: 13.3 ms fp32 / 7.8 ms bf16 at 1M cells), and
// the dense product is 16x cheaper on the matrix cores than on the vector ALUs.  densify.hip + dh_gemm_bf16 does that
// through a 4 GB dense copy of the adjacency; here a workgroup densifies its own 128 cells x 128 genes at a time straight
// into LDS, in MFMA fragment order, and the product never leaves the CU:
//
//   * precision: an adjacency entry a = w * colscale (fp32) is split a = a_hi + a_lo (two bf16, residual 2^-18 |a|); fp32
//     features are split the same way (prepared once per call), bf16 features are exact.  Products: a_hi b_hi + a_hi b_lo +
//     a_lo b_hi (fp32 H; the dropped a_lo b_lo is 2^-18 relative) or a_hi b + a_lo b (bf16 H); each bf16 x bf16 product is
//     exact in fp32 and the matrix core accumulates in fp32.  Worst-case relative error per term 1.2e-5, ~3e-6 of the
//     output's max-norm on the test shapes.
//   * K order: MFMA step j takes k slots 0-7 from genes 8j .. 8j+7 of the LOWER half of the window and slots 8-15 from the
//     same positions of the UPPER half (Gh = half the window, rounded up to 8).  The lane that owns A-fragment row r, k
//     half h (lane = r + 32 h) therefore walks ONE sorted stream — cell r's edges into half h — front to back: 128 lanes,
//     128 streams, no searching inside the loop.  A stream's next 16 (column, weight) pairs are prefetched into registers
//     one chunk ahead; the ones inside the chunk are a prefix (columns ascend), found with compares.
//   * workgroup = 4 wavefronts = 2 cell groups of 64 x 2 column halves; wave (g, n) accumulates 64 cells x up to 7 column
//     tiles of 32 (224 accumulator registers; one wave per SIMD, 512-register budget).  Per chunk of 8 MFMA steps (128
//     genes): zero the group's A image, scatter the streams' entries into it (ds_write_b16 at fragment positions, both
//     planes), then 8 steps of [A fragments 2 x 2 ds_read_b128, per column tile B fragments + 4 or 6 MFMAs]; the prepared
//     feature blocks (K-permuted, fragment order, contiguous 13 / 27 KB per step) arrive by LDS DMA in a three-slot ring.
//
//   * few destination rows (a mini-batch: scDeepSort's reference batch is 500 cells = 4 workgroups, each walking all 125 K-steps: 217 us):
//     the gene window is split over blockIdx.y — up to 256 / row-blocks splits of whole densify chunks —, every split writes its fp32
//     share (scaled by 1 / deg; split 0 adds the out-of-window edges) and sage_mfma_reduce_kernel sums the shares in split order:
//     ~30 us, bit-reproducible; ScDeepSort.fit at batch 500 (100k cells): 0.092 -> 0.067 s per epoch.
//
// Precondition (the layouts CellFeatureGraph and the block builder produce): inside a row the in-window edges are
// contiguous and ascending by column; out-of-window edges sit at the row's ends.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

#ifdef DH_SM_PROF  // development build: cycles per phase (wave 0, s_memtime), summed over the blocks
__device__ unsigned long long dh_sm_prof_cycles[8];
#define PROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define PROF_ADD(i, a, b) do { if (threadIdx.x == 0) prof[i] += (b) - (a); } while (0)
#else
#define PROF_T(x)
#define PROF_ADD(i, a, b)
#endif

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

constexpr int JC = 8;          // MFMA steps per densify chunk (128 window genes: 64 of each half)
constexpr int QD = 16;         // stream entries prefetched per chunk
constexpr int MAX_TILES = 7;   // column tiles of 32 per wave (two waves cover up to 14 tiles = 448 columns)

__device__ __forceinline__ unsigned int f32_to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float widen(unsigned int h) { return __uint_as_float(h << 16); }

// HsP[j][plane][half][Dp][8] bf16 (steps step_stride elements apart: a whole number of 4 KB DMA rounds): k slot s of half h
// of MFMA step j = window gene h * Gh + 8 j + s, column c
template <bool HBF16>
__global__ __launch_bounds__(256) void sage_mfma_prep_kernel(int64_t n_cols, int64_t width, int Gh, int J, int Dp, const void* __restrict__ Hv,
                                                             int64_t ldh, uint16_t* __restrict__ HsP, int64_t step_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (j, h, c)
  if (i >= (int64_t)J * 2 * Dp) return;
  const int c = (int)(i % Dp), h = (int)((i / Dp) % 2), j = (int)(i / (2 * Dp));
  uint16_t hi[8], lo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int64_t g = (int64_t)h * Gh + 8 * j + s;
    const bool live = g < n_cols && c < width && (h == 1 || 8 * j + s < Gh);
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
  uint16_t* o = HsP + j * step_stride + (((int64_t)0 * 2 + h) * Dp + c) * 8;
#pragma unroll
  for (int s = 0; s < 8; ++s) o[s] = hi[s];
  if (!HBF16) {
    uint16_t* o1 = HsP + j * step_stride + (((int64_t)1 * 2 + h) * Dp + c) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) o1[s] = lo[s];
  }
}

// NT = column tiles per wave (both waves of a cell group run NT tile slots; a slot beyond the width computes into an
// accumulator that is never stored, so the step body has no branches and the fragment reads can run ahead of the MFMAs)
template <bool HBF16, bool OBF16, int NT, bool FOLD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sage_mfma_kernel(
    int64_t n_dst, int64_t width, int col_begin, int n_cols, int Gh, int J, int Dp, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ colscale, const uint16_t* __restrict__ HsP,
    void* __restrict__ neigh, int64_t ldn, int64_t nnz, int nbp, const void* __restrict__ Hraw, int64_t ldh,
    const int32_t* __restrict__ src_id, const int32_t* __restrict__ dst_id, const float* __restrict__ alpha, int n_genes,
    float* __restrict__ partial, int chunks_per_split) {
  // partial != null: the gene window is split over blockIdx.y (chunks_per_split densify chunks each) — few destination rows, e.g. a
  // mini-batch of 500 cells, would otherwise occupy ceil(n_dst / 128) of the 256 CUs for the whole 125-step K loop (217 us at 4
  // workgroups).  A split block writes its fp32 share (already scaled by 1 / deg; the out-of-window edges are added by split 0) to
  // partial[blockIdx.y][cell][Dp]; sage_mfma_reduce_kernel sums the shares in split order.
  constexpr int P = HBF16 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // A image: [2 groups][2 planes][JC][2 halves][64 cells][8] bf16 = JC * 8 KB;  B ring: [3 slots][P][2][Dp][8] bf16 (padded);  cs: [n_cols] f32
  uint16_t* const a_img = reinterpret_cast<uint16_t*>(smem);
  constexpr int A_GROUP = 2 * JC * 2 * 64 * 8;  // bf16 elements of one group's image
  uint16_t* const b_img = a_img + 2 * A_GROUP;
  const int b_tile = nbp * 256 * 8;             // bf16 elements of one ring slot = one step's feature block, padded to nbp DMA rounds
  float* const cs = reinterpret_cast<float*>(b_img + 3 * b_tile);
  int* const ends = reinterpret_cast<int*>(cs + n_cols);               // [128 cells][2]: in-window run [s0, e0) of every cell of the block
  uint16_t* const trash = reinterpret_cast<uint16_t*>(ends + 256);   // [256 threads][2]: where the scatter parks entries outside the chunk

#ifdef DH_SM_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PROF_T(t_start);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 1, nh = wave & 1, r = lane & 31, half = lane >> 5;
  const int tiles_total = Dp / 32;
  const int tile0 = nh * NT;
  const int my_tiles = max(0, min(NT, tiles_total - tile0));  // tile slots that hold real columns
  const int64_t cell0 = (int64_t)blockIdx.x * 128 + grp * 64;

  for (int i = tid; i < n_cols; i += 256) cs[i] = colscale ? colscale[i] : 1.f;

  // this lane's stream: the in-window edges of cell (cell0 + 32 nh + r) whose window position falls in half `half`
  const int64_t my_cell = cell0 + 32 * nh + r;
  int p = 0, pend = 0;
  if (my_cell < n_dst) {
    int s0 = rowptr[my_cell], e0 = rowptr[my_cell + 1];
    while (s0 < e0 && (unsigned)(col[s0] - col_begin) >= (unsigned)n_cols) ++s0;      // out-of-window edges sit at the ends
    while (e0 > s0 && (unsigned)(col[e0 - 1] - col_begin) >= (unsigned)n_cols) --e0;
    int lo = s0, hi = e0;  // first edge of the upper half: window position >= Gh
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (col[mid] - col_begin < Gh) lo = mid + 1; else hi = mid;
    }
    p = half ? lo : s0;
    pend = half ? e0 : lo;
    if (partial && blockIdx.y > 0) {  // first edge of this half at or behind the split's first chunk
      const int target = half * Gh + (int)blockIdx.y * chunks_per_split * (JC * 8);
      int a = p, b = pend;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (col[mid] - col_begin < target) a = mid + 1; else b = mid;
      }
      p = a;
    }
    if (half == 0) {
      ends[2 * (grp * 64 + 32 * nh + r)] = s0;
      ends[2 * (grp * 64 + 32 * nh + r) + 1] = e0;
    }
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int y = 0; y < NT; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][y][i] = 0.f;

  // stream prefetch: QD (column, weight) pairs from p (clamped reads: entries at or beyond pend are ignored later)
  int qc[QD];
  float qw[QD];
  auto fetch = [&](int from) __attribute__((always_inline)) {
    if ((int64_t)from + QD <= nnz) {
#pragma unroll
      for (int v = 0; v < QD / 4; ++v) {
        const u32x4_u c4 = *reinterpret_cast<const u32x4_u*>(col + from + 4 * v);
        const u32x4_u w4 = *reinterpret_cast<const u32x4_u*>(w + from + 4 * v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          qc[4 * v + i] = (int)c4[i];
          qw[4 * v + i] = __uint_as_float(w4[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < QD; ++i) {
        const bool ok = (int64_t)from + i < nnz;
        qc[i] = ok ? col[from + i] : 0x7fffffff;
        qw[i] = ok ? w[from + i] : 0.f;
      }
    }
  };
  fetch(p);

  // Feature block of step j: global -> LDS by DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at a wave-uniform LDS
  // base + 16 * lane), into slot j % 3 of a three-slot ring, issued TWO steps ahead; the consumer side waits with a counted
  // vmcnt (the block of step j + 2 stays in flight across the barrier) — with register staging one step ahead every step
  // waited out the L2 latency (10.0 ms at 1M cells instead of the 2.4 ms the MFMAs need).
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  auto b_dma = [&](int j) __attribute__((always_inline)) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(HsP) + ((int64_t)j * b_tile) * 2 + (size_t)tid * 16;
    unsigned char* dst = reinterpret_cast<unsigned char*>(b_img) + ((size_t)(j % 3) * b_tile) * 2 + (size_t)(wave * 64) * 16;  // wave-uniform
#pragma unroll
    for (int s = 0; s < 7; ++s)
      if (s < nbp) __builtin_amdgcn_global_load_lds((glb_void*)(src + s * 4096), (lds_void*)(dst + s * 4096), 16, 0, 0);
  };
  auto wait_vm = [&](int n) __attribute__((always_inline)) {  // s_waitcnt vmcnt(n) needs an immediate
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    }
  };
  const int n_chunks_all = (J + JC - 1) / JC;
  const int chunk_lo = partial ? (int)blockIdx.y * chunks_per_split : 0;
  const int chunk_hi = partial ? min(n_chunks_all, chunk_lo + chunks_per_split) : n_chunks_all;
  const int j_lo = chunk_lo * JC, j_stop = min(J, chunk_hi * JC);  // this block's MFMA steps [j_lo, j_stop)
  if (j_lo < j_stop) b_dma(j_lo);
  if (j_lo + 1 < j_stop) b_dma(j_lo + 1);

  uint16_t* const a_grp = a_img + grp * A_GROUP;
  const int my_cell_local = 32 * nh + r;  // position of my stream's cell inside the group
  __syncthreads();  // the colscale table is complete; the first two feature blocks have landed (vmcnt is drained here)
  PROF_T(t_pro);
  PROF_ADD(0, t_start, t_pro);
  for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
    PROF_T(t_c0);
    // ---- densify this chunk's window genes of the group's 64 cells into fragment order -----------------------------
    {
      // every stream owner clears exactly the 2 planes x JC fragment slots its own entries can land in (its cell, its k half),
      // so no barrier is needed between the clearing and the scatter (LDS executes a lane's accesses in order)
      u32x4* z = reinterpret_cast<u32x4*>(a_grp) + half * 64 + my_cell_local;
#pragma unroll
      for (int i = 0; i < 2 * JC; ++i) z[i * 128] = u32x4(0u);
    }
    PROF_T(t_c1);
    PROF_ADD(1, t_c0, t_c1);
    {
      const int win_lo = half * Gh + chunk * (JC * 8);  // window positions [win_lo, win_hi) of my half belong to this chunk
      const int win_hi = min(win_lo + JC * 8, half ? n_cols : Gh);
#if !defined(DH_SM_ABL) || (DH_SM_ABL != 1 && DH_SM_ABL != 3)
      for (;;) {
        // branch-free: every prefetched entry is converted and written; entries outside the chunk go to a per-thread trash slot
        int n_in = 0;
        float csv[QD];
#pragma unroll
        for (int i = 0; i < QD; ++i) csv[i] = cs[min(max(qc[i] - col_begin, 0), n_cols - 1)];
#pragma unroll
        for (int i = 0; i < QD; ++i) {
          const int g = qc[i] - col_begin;
          const bool in = p + i < pend && g < win_hi;  // ascending columns: a prefix
          n_in += in ? 1 : 0;
          const float a = qw[i] * csv[i];
          const unsigned int ahi = f32_to_bf16(a);
          const unsigned int alo = f32_to_bf16(a - widen(ahi));
          const int gg = g - win_lo;  // 0 .. 63 when `in`
          uint16_t* real = a_grp + ((((gg >> 3) & (JC - 1)) * 2 + half) * 64 + my_cell_local) * 8 + (gg & 7);
          uint16_t* d0 = in ? real : trash + 2 * tid;
          uint16_t* d1 = in ? real + JC * 2 * 64 * 8 : trash + 2 * tid + 1;  // plane 1
          *d0 = (uint16_t)ahi;
          *d1 = (uint16_t)alo;
        }
        p += n_in;
        if (__builtin_expect(n_in < QD, 1)) break;
        fetch(p);  // more than QD entries of this stream in one chunk (3e-4 of the chunks at 10 % density): keep going
      }
#endif
    }
    PROF_T(t_c2);
    PROF_ADD(2, t_c1, t_c2);
    __syncthreads();
    fetch(p);  // next chunk's entries, in flight behind this chunk's MFMAs
    PROF_T(t_c3);
    PROF_ADD(3, t_c2, t_c3);

    // ---- the chunk's MFMA steps ---------------------------------------------------------------------------------------
    const int j_end = min(j_stop, (chunk + 1) * JC);
    bf16x8_t fa[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[pl][m] = *reinterpret_cast<const bf16x8_t*>(a_grp + (((pl * JC + 0) * 2 + half) * 64 + 32 * m + r) * 8);
    for (int j = chunk * JC; j < j_end; ++j) {
      const int jl = j - chunk * JC;
      PROF_T(t_s0);
      if (j + 2 < j_stop) b_dma(j + 2);  // slot (j + 2) % 3 was last read in step j - 1, before that step's barrier
      // tile slots beyond the real columns read a clamped (valid) LDS position and feed accumulators nobody stores
      const uint16_t* bb = b_img + (j % 3) * b_tile + (half * Dp + r) * 8;
      // B fragments in two batches: the second batch's reads are issued before the first batch's MFMAs and have landed when
      // those are through (with LDS DMA in flight the compiler drains lgkmcnt completely at every wait, so a per-tile
      // prefetch exposed the LDS latency per tile; all 4 waves read at once, so the first batch's ~0.3 us is what is left)
      constexpr int NT_A = (NT + 1) / 2;
      bf16x8_t fb[NT][P];
      auto read_b = [&](int y) __attribute__((always_inline)) {
        const int ty = min(tile0 + y, tiles_total - 1);
#pragma unroll
        for (int pl = 0; pl < P; ++pl) fb[y][pl] = *reinterpret_cast<const bf16x8_t*>(bb + (pl * 2 * Dp + ty * 32) * 8);
      };
      auto mma = [&](int y) __attribute__((always_inline)) {  // product planes outermost: consecutive MFMAs alternate accumulators
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[y][0], acc[m][y], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][m], fb[y][0], acc[m][y], 0, 0, 0);
        if (!HBF16) {
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[y][P - 1], acc[m][y], 0, 0, 0);
        }
      };
#pragma unroll
      for (int y = 0; y < NT_A; ++y) read_b(y);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PROF_T(t_s1);
      PROF_ADD(4, t_s0, t_s1);
#pragma unroll
      for (int y = NT_A; y < NT; ++y) read_b(y);
      __builtin_amdgcn_sched_barrier(0);
#if !defined(DH_SM_ABL) || (DH_SM_ABL != 2 && DH_SM_ABL != 3)
#pragma unroll
      for (int y = 0; y < NT_A; ++y) mma(y);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int y = NT_A; y < NT; ++y) mma(y);
#else
#pragma unroll
      for (int y = 0; y < NT; ++y) acc[0][y][0] += (float)fb[y][0][0] + (float)fa[0][0][0];  // ablation build: keep the reads alive
#endif
      __builtin_amdgcn_sched_barrier(0);
      // the A fragments of the next step of this chunk come from the same (stable) image: fetch them across the barrier
      if (j + 1 < j_end) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            fa[pl][m] = *reinterpret_cast<const bf16x8_t*>(a_grp + (((pl * JC + jl + 1) * 2 + half) * 64 + 32 * m + r) * 8);
      }
      // the block of step j + 1 has landed for THIS wave once at most the nbp DMA instructions of step j + 2 are outstanding
      // (any younger register load only makes the wait longer); then the workgroup meets, without draining vmcnt
      PROF_T(t_s2);
      PROF_ADD(5, t_s1, t_s2);
      wait_vm(j + 2 < j_stop ? nbp : 0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PROF_T(t_s3);
      PROF_ADD(6, t_s2, t_s3);
    }
  }

  PROF_T(t_epi);
  // ---- epilogue: neigh[cell, col] += acc / deg(cell).  C layout: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5):
  // every accumulator tile goes through a private LDS patch [32][36] so that a lane then owns 16 CONSECUTIVE columns of one
  // cell and the read-modify-write runs on 16-byte accesses (element-wise it was a third of the kernel: 224 dependent 4-byte
  // load / store pairs per lane).
  __syncthreads();  // the images are dead: reuse their space
  {
    float* T = reinterpret_cast<float*>(smem) + wave * (32 * 36);
    const int rr = lane >> 1, c16 = (lane & 1) * 16;
    // (compile-time loops over the accumulator tiles: with `#pragma unroll` the largest instantiations were left rolled and
    // their accumulators went to scratch)
    static_for<2>([&](auto m_c) __attribute__((always_inline)) {
      constexpr int m = decltype(m_c)::value;
      const int64_t cell = cell0 + 32 * m + rr;
      float inv = 0.f;
      // FOLD: the in-edges OUTSIDE the window (the self loop; they sit at the row's ends: [rs, s0) and [e0, re)) are added
      // here with the alpha rule of gnn.py:72-76, and neigh is written once instead of read-modify-written
      int n_tail = 0, t_skip = 0, tck[4] = {0, 0, 0, 0};
      float tfk[4] = {0.f, 0.f, 0.f, 0.f};
      if (cell < n_dst) {
        const int rs = rowptr[cell], re = rowptr[cell + 1];
        const int d = re - rs;
        inv = d > 0 ? 1.f / (float)d : 0.f;
        if (FOLD) {
          const int s0 = ends[2 * (grp * 64 + 32 * m + rr)], e0 = ends[2 * (grp * 64 + 32 * m + rr) + 1];
          n_tail = (partial && blockIdx.y != 0) ? 0 : (s0 - rs) + (re - e0);  // split blocks: the out-of-window edges belong to split 0
          t_skip = e0 - s0;  // tail edge q lives at rs + q (q < s0 - rs) or rs + q + (e0 - s0)
          const int did = dst_id[cell];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < n_tail) {
              const int e = rs + q + (q < s0 - rs ? 0 : t_skip);
              const int u = col[e];
              const int sid = src_id[u];
              int idx = n_genes + 1;
              if (sid >= 0 && did < 0) idx = sid;
              if (did >= 0 && sid < 0) idx = did;
              if (did >= 0 && sid >= 0) idx = n_genes;
              tck[q] = u;
              tfk[q] = w[e] * alpha[idx];
            }
        }
      }
      auto h_row16 = [&](int u, int64_t c0, float (&hv)[16]) __attribute__((always_inline)) {  // H[u][c0 .. c0 + 15] (zero beyond width)
        if (HBF16) {
          const uint16_t* hp = static_cast<const uint16_t*>(Hraw) + (int64_t)u * ldh + c0;
          if (c0 + 16 <= width) {
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
              const u32x4_u t4 = *reinterpret_cast<const u32x4_u*>(hp + 8 * q4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                hv[8 * q4 + 2 * e] = __uint_as_float(t4[e] << 16);
                hv[8 * q4 + 2 * e + 1] = __uint_as_float(t4[e] & 0xffff0000u);
              }
            }
          } else {
            const int last = (int)(width - 1 - c0);  // >= 0: the caller checked c0 < width
#pragma unroll
            for (int e = 0; e < 16; ++e) hv[e] = e <= last ? widen(hp[min(e, last)]) : 0.f;
          }
        } else {
          const float* hp = static_cast<const float*>(Hraw) + (int64_t)u * ldh + c0;
          if (c0 + 16 <= width) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const f32x4_u t4 = *reinterpret_cast<const f32x4_u*>(hp + 4 * q4);
#pragma unroll
              for (int e = 0; e < 4; ++e) hv[4 * q4 + e] = t4[e];
            }
          } else {
            const int last = (int)(width - 1 - c0);
#pragma unroll
            for (int e = 0; e < 16; ++e) hv[e] = e <= last ? hp[min(e, last)] : 0.f;
          }
        }
      };
      // the first out-of-window edge's feature segments (the self loop's row of H: cold in the caches) are requested one tile
      // ahead of their use
      float hvp[2][16];
      auto pre_ok = [&](int yy) -> bool { return FOLD && n_tail > 0 && cell < n_dst && yy < my_tiles && (int64_t)(tile0 + yy) * 32 + c16 < width; };
      if (pre_ok(0)) h_row16(tck[0], (int64_t)tile0 * 32 + c16, hvp[0]);
      static_for<NT>([&](auto y_c) __attribute__((always_inline)) {
        constexpr int y = decltype(y_c)::value;
        if (y < my_tiles) {  // wave-uniform
        const bool pre = pre_ok(y);
        float (&hv0)[16] = hvp[y & 1];
        if (y + 1 < NT && pre_ok(y + 1)) h_row16(tck[0], (int64_t)(tile0 + y + 1) * 32 + c16, hvp[(y + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 16; ++i) T[((i & 3) + 8 * (i >> 2) + 4 * half) * 36 + r] = acc[m][y][i];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(T + rr * 36 + c16 + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = t4[e];
        }
        __builtin_amdgcn_wave_barrier();
        const int64_t c0 = (int64_t)(tile0 + y) * 32 + c16;
        if (FOLD && cell < n_dst && c0 < width) {
          float hv[16];
          if (pre) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = fmaf(tfk[0], hv0[e], v[e]);
          }
#pragma unroll
          for (int q = 1; q < 4; ++q)
            if (q < n_tail) {
              h_row16(tck[q], c0, hv);
#pragma unroll
              for (int e = 0; e < 16; ++e) v[e] = fmaf(tfk[q], hv[e], v[e]);
            }
          if (partial) {  // this split's share, fp32, rows Dp wide (16-byte aligned for every c0)
            float* o = partial + ((int64_t)blockIdx.y * n_dst + cell) * Dp + c0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              f32x4_t t4;
#pragma unroll
              for (int e = 0; e < 4; ++e) t4[e] = v[4 * q4 + e] * inv;
              *reinterpret_cast<f32x4_t*>(o + 4 * q4) = t4;
            }
          } else if (OBF16) {
            uint16_t* o = static_cast<uint16_t*>(neigh) + cell * ldn + c0;
            if (c0 + 16 <= width && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0)) {
#pragma unroll
              for (int q4 = 0; q4 < 2; ++q4) {
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = f32_to_bf16(v[8 * q4 + 2 * e] * inv) | (f32_to_bf16(v[8 * q4 + 2 * e + 1] * inv) << 16);
                *reinterpret_cast<u32x4*>(o + 8 * q4) = pk;
              }
            } else {
              const int last = (int)(width - 1 - c0);
#pragma unroll
              for (int e = 0; e < 16; ++e)
                if (e <= last) o[e] = (uint16_t)f32_to_bf16(v[e] * inv);
            }
          } else {
            float* o = static_cast<float*>(neigh) + cell * ldn + c0;
            if (c0 + 16 <= width && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0)) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                f32x4_t t4;
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = v[4 * q4 + e] * inv;
                *reinterpret_cast<f32x4_t*>(o + 4 * q4) = t4;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e)
                if (c0 + e < width) o[e] = v[e] * inv;
            }
          }
        } else if (!FOLD && cell < n_dst && c0 < width) {
          if (partial) {
            float* o = partial + ((int64_t)blockIdx.y * n_dst + cell) * Dp + c0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              f32x4_t t4;
#pragma unroll
              for (int e = 0; e < 4; ++e) t4[e] = v[4 * q4 + e] * inv;
              *reinterpret_cast<f32x4_t*>(o + 4 * q4) = t4;
            }
          } else if (OBF16) {
            uint16_t* o = static_cast<uint16_t*>(neigh) + cell * ldn + c0;
            if (c0 + 16 <= width && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0)) {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                u32x4 pk = *reinterpret_cast<const u32x4*>(o + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float lo = fmaf(v[8 * q + 2 * e], inv, __uint_as_float(pk[e] << 16));
                  const float hi = fmaf(v[8 * q + 2 * e + 1], inv, __uint_as_float(pk[e] & 0xffff0000u));
                  pk[e] = f32_to_bf16(lo) | (f32_to_bf16(hi) << 16);
                }
                *reinterpret_cast<u32x4*>(o + 8 * q) = pk;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e)
                if (c0 + e < width) o[e] = (uint16_t)f32_to_bf16(fmaf(v[e], inv, widen(o[e])));
            }
          } else {
            float* o = static_cast<float*>(neigh) + cell * ldn + c0;
            if (c0 + 16 <= width && ((reinterpret_cast<uintptr_t>(o) & 15u) == 0)) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(o + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = fmaf(v[4 * q + e], inv, t4[e]);
                *reinterpret_cast<f32x4_t*>(o + 4 * q) = t4;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 16; ++e)
                if (c0 + e < width) o[e] = fmaf(v[e], inv, o[e]);
            }
          }
        }
        }
      });
    });
  }
  if (FOLD) {
    // rows with more than four out-of-window edges (not a CellFeatureGraph row, but legal): the rest is added to the
    // written result element-wise (read-modify-write after the block's own stores; only this block touches these rows)
    __syncthreads();
    const int rr = lane >> 1;
    for (int m = 0; m < 2; ++m) {
      const int64_t cell = cell0 + 32 * m + rr;
      if (cell >= n_dst) continue;
      const int rs = rowptr[cell], re = rowptr[cell + 1];
      const int s0 = ends[2 * (grp * 64 + 32 * m + rr)], e0 = ends[2 * (grp * 64 + 32 * m + rr) + 1];
      const int n_tail = (s0 - rs) + (re - e0);
      if (n_tail <= 4 || nh != 0 || (partial && blockIdx.y != 0)) continue;  // one wave per cell group (of split 0) does the whole row
      const float inv = 1.f / (float)(re - rs);
      const int did = dst_id[cell];
      for (int q = 4; q < n_tail; ++q) {
        const int e1 = rs + q + (q < s0 - rs ? 0 : e0 - s0);
        const int u = col[e1];
        const int sid = src_id[u];
        int idx = n_genes + 1;
        if (sid >= 0 && did < 0) idx = sid;
        if (did >= 0 && sid < 0) idx = did;
        if (did >= 0 && sid >= 0) idx = n_genes;
        const float f = w[e1] * alpha[idx] * inv;
        for (int64_t c = (lane & 1); c < width; c += 2) {
          const float hvv = HBF16 ? widen(static_cast<const uint16_t*>(Hraw)[(int64_t)u * ldh + c]) : static_cast<const float*>(Hraw)[(int64_t)u * ldh + c];
          if (partial) {
            float* o = partial + cell * Dp + c;  // split 0's share
            *o = fmaf(f, hvv, *o);
          } else if (OBF16) {
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
#ifdef DH_SM_PROF
  {
    PROF_T(t_end);
    PROF_ADD(7, t_epi, t_end);
    if (threadIdx.x == 0)
      for (int i = 0; i < 8; ++i) atomicAdd(&dh_sm_prof_cycles[i], prof[i]);
  }
#endif
}

// neigh[cell][c] (+)= sum over the splits, in split order (deterministic); accumulate = the non-folded form adds to what dh_sage_tail wrote
template <bool OBF16>
__global__ __launch_bounds__(256) void sage_mfma_reduce_kernel(int64_t n_dst, int64_t width, int Dp, int S, const float* __restrict__ partial,
                                                               void* __restrict__ neigh, int64_t ldn, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_dst * Dp) return;
  const int64_t cell = i / Dp;
  const int c = (int)(i - cell * Dp);
  if (c >= width) return;
  float v = 0.f;
  {
      int s = 0;
      for (; s + 8 <= S; s += 8) {  // eight partial values in flight, added in order (a plain loop is S dependent round trips)
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = partial[((int64_t)(s + u) * n_dst + cell) * Dp + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t8[u];
      }
      for (; s < S; ++s) v += partial[((int64_t)s * n_dst + cell) * Dp + c];
    }
  if (OBF16) {
    uint16_t* o = static_cast<uint16_t*>(neigh) + cell * ldn + c;
    *o = (uint16_t)f32_to_bf16(accumulate ? v + widen(*o) : v);
  } else {
    float* o = static_cast<float*>(neigh) + cell * ldn + c;
    *o = accumulate ? v + *o : v;
  }
}

struct Geo {
  int Gh, J, Dp, nbp;
  size_t prep_bytes, lds_bytes;
};
Geo geometry(int64_t n_cols, int64_t width, bool hbf16) {
  Geo g;
  g.Gh = (int)((((n_cols + 1) / 2) + 7) / 8 * 8);
  g.J = g.Gh / 8;
  g.Dp = (int)((width + 31) / 32 * 32);
  const int P = hbf16 ? 1 : 2;
  g.nbp = (P * 2 * g.Dp * 16 + 4095) / 4096;  // DMA rounds (256 threads x 16 B) per step's feature block
  g.prep_bytes = (size_t)g.J * g.nbp * 4096;
  g.lds_bytes = (size_t)2 * (2 * JC * 2 * 64 * 8) * 2 + (size_t)3 * g.nbp * 4096 + (size_t)n_cols * 4 + 256 * 4 + 256 * 4;
  return g;
}

}  // namespace

#ifdef DH_SM_PROF
extern "C" __attribute__((visibility("default"))) int dh_sage_mfma_prof_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(dh_sm_prof_cycles), 64) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dh_sm_prof_cycles), z, 64) != hipSuccess) return -1;
  }
  return 0;
}
#endif

extern "C" int dh_sage_window_mfma_supported(int64_t n_cols, int64_t width, int h_dtype) {
  if (n_cols <= 0 || width <= 0 || width > 32 * 2 * MAX_TILES || n_cols > 4096) return 0;
  return geometry(n_cols, width, h_dtype == DH_DTYPE_BF16).lds_bytes <= 160 * 1024 ? 1 : 0;
}

extern "C" size_t dh_sage_window_mfma_workspace_bytes(int64_t n_cols, int64_t width, int h_dtype) {
  if (n_cols <= 0 || width <= 0) return 0;
  return geometry(n_cols, width, h_dtype == DH_DTYPE_BF16).prep_bytes;
}

namespace {
// how many ways the gene window is split for a launch of n_dst rows: 1 when the row blocks alone cover the 256 CUs, else enough splits
// for about two workgroups per CU
int split_factor(int64_t n_dst, const Geo& g) {
  const int64_t blocks = dh::ceil_div(n_dst, 128);
  const int n_chunks = (g.J + JC - 1) / JC;
  if (blocks >= 256 || n_chunks < 2) return 1;
  const int64_t want = dh::ceil_div((int64_t)512, blocks);
  return (int)(want < n_chunks ? want : n_chunks);
}
}  // namespace

/* prep buffer + the fp32 shares of a split launch: hand this many bytes to dh_sage_window_mfma and few-row launches (mini-batches) split
 * the gene window over the otherwise idle CUs; with only dh_sage_window_mfma_workspace_bytes the unsplit kernel runs. */
extern "C" size_t dh_sage_window_mfma_split_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype) {
  if (n_dst <= 0 || n_cols <= 0 || width <= 0) return 0;
  const Geo g = geometry(n_cols, width, h_dtype == DH_DTYPE_BF16);
  const int S = split_factor(n_dst, g);
  const size_t prep = (g.prep_bytes + 255) / 256 * 256;
  return S > 1 ? prep + (size_t)S * n_dst * g.Dp * sizeof(float) : g.prep_bytes;
}

/* workspace of the repack + two-waves-per-SIMD kernel pair (sage_bcm.hip) that unsplit launches run when it is provided: the K-major feature
 * planes, the chunk pointers of every 64-row group and the packed entries (8 bytes per stored entry of col / w); 0 if the shape does not fit. */
extern "C" size_t dh_sage_window_mfma_bcm_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype, int64_t nnz) {
  if (n_dst <= 0 || n_cols <= 0 || width <= 0 || nnz <= 0 || width > 512 || n_cols > 4096) return 0;
  return dh::sage_bcm_workspace_bytes(n_dst, n_cols, width, h_dtype == DH_DTYPE_BF16, nnz);
}

extern "C" int dh_sage_window_mfma(int64_t n_dst, int64_t n_src, int64_t width, int64_t col_begin, int64_t n_cols,
                                   const int32_t* rowptr, const int32_t* col, const float* w, const float* colscale, const void* H,
                                   int64_t ldh, int h_dtype, void* neigh, int64_t ldn, int out_dtype, int64_t nnz,
                                   const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha, int64_t n_genes,
                                   void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_sage_window_mfma";
  if (n_dst < 0 || n_src < 0 || width < 0 || col_begin < 0 || n_cols < 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_dst == 0 || width == 0 || n_cols == 0) return DH_OK;
  if (!rowptr || !col || !w || !H || !neigh) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (ldh < width || ldn < width) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < width", me);
  if ((h_dtype != DH_DTYPE_F32 && h_dtype != DH_DTYPE_BF16) || (out_dtype != DH_DTYPE_F32 && out_dtype != DH_DTYPE_BF16)) return dh::fail(DH_ERR_INVALID, "%s: bad dtype", me);
  if (col_begin + n_cols > n_src) return dh::fail(DH_ERR_INVALID, "%s: window beyond the source rows", me);
  if (width > 32 * 2 * MAX_TILES) return dh::fail(DH_ERR_INVALID, "%s: width %lld > %d", me, (long long)width, 32 * 2 * MAX_TILES);
  if (n_cols > 4096) return dh::fail(DH_ERR_INVALID, "%s: window of %lld columns > 4096", me, (long long)n_cols);
  const bool fold = src_cell_id || dst_cell_id || alpha;
  if (fold && !(src_cell_id && dst_cell_id && alpha)) return dh::fail(DH_ERR_INVALID, "%s: src_cell_id, dst_cell_id and alpha go together", me);
  const bool hb = h_dtype == DH_DTYPE_BF16, ob = out_dtype == DH_DTYPE_BF16;
  const Geo g = geometry(n_cols, width, hb);
  if (!workspace || workspace_bytes < g.prep_bytes) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, workspace_bytes, g.prep_bytes);
  if (g.lds_bytes > 160 * 1024) return dh::fail(DH_ERR_INVALID, "%s: needs %zu bytes of LDS (window too wide for this width)", me, g.lds_bytes);
  hipStream_t st = dh::as_stream(stream);
  {
    // Unsplit launches (>= 256 row blocks: the full graph, large batches) with the self loops folded in run the repack + two-waves-
    // per-SIMD kernel pair of sage_bcm.hip when the caller provided its workspace (dh_sage_window_mfma_bcm_workspace_bytes).
    // DANCE_AMD_SAGE_MFMA = "v1" keeps this file's kernel everywhere (A/B), "bcm" forces the new pair for few-row launches too
    // (tests: every shape through both).
    const char* mode = getenv("DANCE_AMD_SAGE_MFMA");
    const bool v1_only = mode && mode[0] == 'v' && mode[1] == '1';
    const bool bcm_force = mode && mode[0] == 'b';
    const int s_try = split_factor(n_dst, g);
    const bool unsplit = s_try == 1 || workspace_bytes < (g.prep_bytes + 255) / 256 * 256 + (size_t)s_try * n_dst * g.Dp * sizeof(float);
    if (fold && !v1_only && (unsplit || bcm_force) && dh::sage_bcm_fits(n_dst, n_cols, width, hb, H, ldh, nnz) &&
        workspace_bytes >= dh::sage_bcm_workspace_bytes(n_dst, n_cols, width, hb, nnz))
    {  // the plan (graph only) behind the feature planes in the caller's workspace, rebuilt per call; callers that keep a graph use
       // dh_sage_window_plan once and dh_sage_window_mfma_planned per call
      char* plan = static_cast<char*>(workspace) + dh::sage_bcm_prep_bytes(n_cols, width, hb);
      if (int rc = dh::sage_bcm_plan(n_dst, col_begin, n_cols, rowptr, col, w, plan, st)) return rc;
      return dh::sage_bcm_launch(n_dst, width, col_begin, n_cols, rowptr, col, w, colscale, H, ldh, hb, neigh, ldn, ob, nnz, src_cell_id,
                                 dst_cell_id, alpha, n_genes, plan, workspace, st);
    }
  }
  uint16_t* HsP = static_cast<uint16_t*>(workspace);
  const char* Hw = static_cast<const char*>(H) + (size_t)col_begin * ldh * (hb ? 2 : 4);
  const unsigned pgrid = (unsigned)dh::ceil_div((int64_t)g.J * 2 * g.Dp, 256);
  const int64_t step_stride = (int64_t)g.nbp * 2048;  // bf16 elements
  if (hb) hipLaunchKernelGGL(sage_mfma_prep_kernel<true>, dim3(pgrid), dim3(256), 0, st, n_cols, width, g.Gh, g.J, g.Dp, Hw, ldh, HsP, step_stride);
  else hipLaunchKernelGGL(sage_mfma_prep_kernel<false>, dim3(pgrid), dim3(256), 0, st, n_cols, width, g.Gh, g.J, g.Dp, Hw, ldh, HsP, step_stride);
  const unsigned grid = (unsigned)dh::ceil_div(n_dst, 128);
  int S = split_factor(n_dst, g);
  const size_t prep_aligned = (g.prep_bytes + 255) / 256 * 256;
  if (S > 1 && workspace_bytes < prep_aligned + (size_t)S * n_dst * g.Dp * sizeof(float)) S = 1;  // caller did not provide room for the shares
  float* partial = S > 1 ? reinterpret_cast<float*>(static_cast<char*>(workspace) + prep_aligned) : nullptr;
  const int n_chunks = (g.J + JC - 1) / JC;
  const int cps = S > 1 ? (n_chunks + S - 1) / S : n_chunks;
  if (S > 1) S = (n_chunks + cps - 1) / cps;  // no empty split
  const int tiles_total = g.Dp / 32;
  const int nt = tiles_total <= 4 ? 2 : tiles_total <= 8 ? 4 : MAX_TILES;
#define DH_SM4(HB, OB, NTV, FD)                                                                                                    \
  do {                                                                                                                             \
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(sage_mfma_kernel<HB, OB, NTV, FD>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;              \
    if (!ok) return dh::fail(DH_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit", me);                                         \
    hipLaunchKernelGGL((sage_mfma_kernel<HB, OB, NTV, FD>), dim3(grid, (unsigned)S), dim3(256), g.lds_bytes, st, n_dst, width,     \
                       (int)col_begin, (int)n_cols, g.Gh, g.J, g.Dp, rowptr, col, w, colscale, HsP, neigh, ldn, nnz, g.nbp, H, ldh, \
                       src_cell_id, dst_cell_id, alpha, (int)n_genes, partial, cps);                                               \
  } while (0)
#define DH_SM3(HB, OB, NTV)                                                                                                        \
  do {                                                                                                                             \
    if (fold) DH_SM4(HB, OB, NTV, true);                                                                                           \
    else DH_SM4(HB, OB, NTV, false);                                                                                               \
  } while (0)
#define DH_SM(HB, OB)                                                                                                              \
  do {                                                                                                                             \
    if (nt == 2) DH_SM3(HB, OB, 2);                                                                                                \
    else if (nt == 4) DH_SM3(HB, OB, 4);                                                                                           \
    else DH_SM3(HB, OB, MAX_TILES);                                                                                                \
  } while (0)
  if (hb && ob) DH_SM(true, true);
  else if (hb) DH_SM(true, false);
  else if (ob) DH_SM(false, true);
  else DH_SM(false, false);
#undef DH_SM
#undef DH_SM3
#undef DH_SM4
  int rc = dh::check_launch(me);
  if (rc != DH_OK || S == 1) return rc;
  const unsigned rgrid = (unsigned)dh::ceil_div(n_dst * g.Dp, 256);
  if (ob) hipLaunchKernelGGL(sage_mfma_reduce_kernel<true>, dim3(rgrid), dim3(256), 0, st, n_dst, width, g.Dp, S, partial, neigh, ldn, fold ? 0 : 1);
  else hipLaunchKernelGGL(sage_mfma_reduce_kernel<false>, dim3(rgrid), dim3(256), 0, st, n_dst, width, g.Dp, S, partial, neigh, ldn, fold ? 0 : 1);
  return dh::check_launch(me);
}

/* ---- the plan interface of the two-waves-per-SIMD path (sage_bcm.hip) ---------------------------------------------------------------- */
extern "C" size_t dh_sage_window_plan_bytes(int64_t n_dst, int64_t n_cols, int64_t nnz) {
  if (n_dst <= 0 || n_cols <= 0 || nnz <= 0 || n_cols > 4096) return 0;
  return dh::sage_bcm_plan_bytes(n_dst, n_cols, nnz);
}

extern "C" int dh_sage_window_plan(int64_t n_dst, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col, const float* w,
                                   int64_t nnz, void* plan, size_t plan_bytes, dh_stream_t stream) {
  const char* me = "dh_sage_window_plan";
  if (n_dst < 0 || col_begin < 0 || n_cols <= 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "%s: bad size", me);
  if (n_dst == 0) return DH_OK;
  if (!rowptr || !col || !w || !plan) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (n_cols > 4096 || nnz < 2 || nnz >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: window of %lld columns / %lld entries not supported", me, (long long)n_cols, (long long)nnz);
  if (plan_bytes < dh::sage_bcm_plan_bytes(n_dst, n_cols, nnz)) return dh::fail(DH_ERR_WORKSPACE, "%s: plan buffer %zu < %zu bytes", me, plan_bytes, dh::sage_bcm_plan_bytes(n_dst, n_cols, nnz));
  return dh::sage_bcm_plan(n_dst, col_begin, n_cols, rowptr, col, w, plan, dh::as_stream(stream));
}

extern "C" int dh_sage_window_mfma_planned_supported(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype, const void* H, int64_t ldh, int64_t nnz) {
  return dh::sage_bcm_fits(n_dst, n_cols, width, h_dtype == DH_DTYPE_BF16, H, ldh, nnz) ? 1 : 0;
}

extern "C" int dh_sage_window_mfma_planned(int64_t n_dst, int64_t n_src, int64_t width, int64_t col_begin, int64_t n_cols, const int32_t* rowptr,
                                           const int32_t* col, const float* w, const float* colscale, const void* H, int64_t ldh, int h_dtype,
                                           void* neigh, int64_t ldn, int out_dtype, int64_t nnz, const int32_t* src_cell_id,
                                           const int32_t* dst_cell_id, const float* alpha, int64_t n_genes, const void* plan, size_t plan_bytes,
                                           void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_sage_window_mfma_planned";
  if (n_dst < 0 || n_src < 0 || width < 0 || col_begin < 0 || n_cols < 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n_dst == 0 || width == 0 || n_cols == 0) return DH_OK;
  if (!rowptr || !col || !w || !H || !neigh || !plan || !workspace) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (!src_cell_id || !dst_cell_id || !alpha) return dh::fail(DH_ERR_INVALID, "%s: src_cell_id, dst_cell_id and alpha are required (the self loops are folded in)", me);
  if (ldh < width || ldn < width) return dh::fail(DH_ERR_INVALID, "%s: leading dimension < width", me);
  if ((h_dtype != DH_DTYPE_F32 && h_dtype != DH_DTYPE_BF16) || (out_dtype != DH_DTYPE_F32 && out_dtype != DH_DTYPE_BF16)) return dh::fail(DH_ERR_INVALID, "%s: bad dtype", me);
  if (col_begin + n_cols > n_src) return dh::fail(DH_ERR_INVALID, "%s: window beyond the source rows", me);
  const bool hb = h_dtype == DH_DTYPE_BF16;
  if (!dh::sage_bcm_fits(n_dst, n_cols, width, hb, H, ldh, nnz)) return dh::fail(DH_ERR_INVALID, "%s: shape / alignment not supported (see dh_sage_window_mfma_planned_supported)", me);
  if (plan_bytes < dh::sage_bcm_plan_bytes(n_dst, n_cols, nnz)) return dh::fail(DH_ERR_WORKSPACE, "%s: plan buffer %zu < %zu bytes", me, plan_bytes, dh::sage_bcm_plan_bytes(n_dst, n_cols, nnz));
  if (workspace_bytes < dh::sage_bcm_prep_bytes(n_cols, width, hb)) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, workspace_bytes, dh::sage_bcm_prep_bytes(n_cols, width, hb));
  return dh::sage_bcm_launch(n_dst, width, col_begin, n_cols, rowptr, col, w, colscale, H, ldh, hb, neigh, ldn, out_dtype == DH_DTYPE_BF16, nnz,
                             src_cell_id, dst_cell_id, alpha, n_genes, plan, workspace, dh::as_stream(stream));
}

extern "C" size_t dh_sage_window_mfma_planned_workspace_bytes(int64_t n_cols, int64_t width, int h_dtype) {
  if (n_cols <= 0 || width <= 0) return 0;
  return dh::sage_bcm_prep_bytes(n_cols, width, h_dtype == DH_DTYPE_BF16);
}
