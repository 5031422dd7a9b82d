// The persistent mini-batch training steps of graph-sc and scDeepSort at the reference's batch sizes (graphsc.py:181-230, batch 128;
// scdeepsort.py:222-257, batch 500): one C call runs a whole run of steps, FOUR kernel launches per step, nothing else on the stream.
//
// Why: with one framework op per reference line a step was 66 launches of ~5 us (block build + degree scalings + a transposed copy
// of the block through three rocPRIM sorts / scans, seven small GEMMs, dropout, decoder, torch's Adam): 0.29 ms per graph-sc step,
// 0.24 ms per scDeepSort step, launch-bound at ~0.05 of any roofline (profiles/r05u_graphsc_step_kernels_b128.md).  What the step
// needs is much less than what it was given:
//   * no block.  A seed CELL's row of a CellFeatureGraph-layout CSR already IS its block row (its genes ascending, then its self
//     loop): the kernels walk the parent graph's rows; dgl.to_block's column renumbering exists only to index a gathered feature copy,
//     and the gene rows of the feature matrix are the same for every batch.
//   * no transpose.  graph-sc's input features carry no gradient (graphsc.py:200: a leaf), and the layer is linear in them, so with
//     the aggregation done FIRST — AX = A_norm X at width in_feats = 50, then (AX) W1 — the backward is dW1 = AX^T dPre: a dense
//     product over the batch rows; A^T never appears.  (The reference multiplies by W first, graphsc.py:452-465; SURVEY 8(d) allows
//     either order when stated — this is the aggregate-first order; results agree to fp32 rounding, tests pin both to the goldens.)
//   * weight gradients + bias gradients + Adam are ONE kernel: 32 x 32 tiles of A^T B on the fp32 matrix cores (exact fp32,
//     v_mfma_f32_32x32x2_f32), each finished tile applying torch.optim.Adam's update to its 1024 parameters in the epilogue.
//   * dropout is drawn in-kernel: Philox4x32-10 keyed by (seed, step, stream, element), the seed taken from torch's generator by the
//     host (so torch.manual_seed reproduces a run); masks are never materialised except for the shared gene rows.
//
// graph-sc step (B seeds, G genes, F -> H -> E; two forwards per batch as graphsc.py:202,215 writes it):
//   gsc_prepare : per-gene out-degree of the batch (int atomics), layout checks, dropout of the G x F gene rows for both forwards,
//                 Adam step counters + bias corrections
//   gsc_forward : one workgroup per seed, both forwards: AX (row gather over the CSR row, D_out^-1/2 per source, D_in^-1/2), ReLU(AX W1 + b1),
//                 Linear; forward 0 writes the epoch's embedding, forward 1 keeps AX, h and the decoder-dropped embedding
//   gsc_decoder : one workgroup per seed: its row of z z^T, weighted BCE against the identity target (the only cell -> cell edges of a
//                 batch are the self loops; checked by gsc_prepare), d/dz, decoder-dropout backward, Linear backward, ReLU mask
//   ms_grad     : dW1, db1, dW2, db2 tiles + Adam; the step's loss; the gene counters zeroed for the next step
// scDeepSort step (B seeds, D -> H -> C classes; the layer output ignores the aggregation, as the reference's does: gnn.py:92-96):
//   sds_neigh   : (optional: AdaptiveSAGE.compute_neigh) the discarded weighted mean over the seed rows, straight off the CSR
//   sds_hidden  : relu(dropout(feat[seeds]) W1^T + b1), 32 x 32 matrix-core tiles over GATHERED feature rows (fp32 or bf16 storage)
//   sds_loss    : logits, summed cross entropy, softmax - onehot, its product with W2 masked by the ReLU: one wavefront per seed
//   ms_grad     : dW1 (over the gathered rows again), db1, dW2, db2 + Adam; the step's loss
//
// Deterministic: fixed summation orders everywhere (integer atomics only).  No memset / memcpy nodes (capturable), no host reads.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's CUDA dropout) ------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r) {
      k.x += 0x9E3779B9u;
      k.y += 0xBB67AE85u;
    }
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
  }
  return c;
}

// one dropout layer's draw for one step: element e of stream `sid` is kept iff the top 24 bits of word (e & 3) of
// philox(counter = (e >> 2, sid, step_lo, step_hi), key = seed) are below thr = floor((1 - p) 2^24); kept values are scaled by 1 / (1 - p)
struct Drop {
  uint32_t seed_lo, seed_hi, step_lo, step_hi, thr;
  float scale;
};
constexpr uint32_t kKeepAll = 1u << 24;

__device__ __forceinline__ uint4 drop_words(const Drop& d, uint32_t sid, uint64_t quad) {
  return philox4x32_10(make_uint4((uint32_t)quad, sid | ((uint32_t)(quad >> 32) << 8), d.step_lo, d.step_hi), make_uint2(d.seed_lo, d.seed_hi));
}
__device__ __forceinline__ float drop_pick(const Drop& d, const uint4& r, int j) {
  const uint32_t w = j == 0 ? r.x : j == 1 ? r.y : j == 2 ? r.z : r.w;
  return (w >> 8) < d.thr ? d.scale : 0.f;
}
__device__ __forceinline__ float drop_scale(const Drop& d, uint32_t sid, uint64_t e) {
  if (d.thr >= kKeepAll) return 1.f;
  const uint4 r = drop_words(d, sid, e >> 2);
  return drop_pick(d, r, (int)(e & 3));
}

// stream ids
constexpr uint32_t SID_GENE = 0;  // + forward: the G x F gene rows
constexpr uint32_t SID_SELF = 2;  // + forward: the B x F seed rows
constexpr uint32_t SID_DEC = 4;   // the decoder's B x E draw
constexpr uint32_t SID_SDS = 5;   // scDeepSort: the B x D seed rows

__host__ __device__ inline int pow2_at_least(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// sum_k v[k] * g[k * stride] for k < n, U global loads in flight per thread (v in LDS / registers-by-broadcast).  A dependent round trip
// to L2 / MALL costs ~1 us in these one-workgroup-per-CU kernels (every step rewrites weights and activations, so a kernel's first touch
// of a line is never an L1 / own-L2 hit): the chains are kept SHORT — n / U rounds — rather than the loads few.  Clamped addresses, zero
// weights past n: no serial remainder loop.
template <int U>
__device__ __forceinline__ float dot_strided(const float* v, const float* __restrict__ g, int64_t stride, int n) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < n; k0 += U) {
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = g[(int64_t)min(k0 + u, n - 1) * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u & 3] = fmaf(v[min(k0 + u, n - 1)], k0 + u < n ? w[u] : 0.f, acc[u & 3]);  // (a select on the VALUE: a guarded LDS read is a branch)
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// two vectors against the same strided column (both forwards of a seed share the weight loads)
template <int U>
__device__ __forceinline__ void dot2_strided(const float* v0, const float* v1, const float* __restrict__ g, int64_t stride, int n, float& r0, float& r1) {
  float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
  for (int k0 = 0; k0 < n; k0 += U) {
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = g[(int64_t)min(k0 + u, n - 1) * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kc = min(k0 + u, n - 1);
      const float wu = k0 + u < n ? w[u] : 0.f;
      a0[u & 1] = fmaf(v0[kc], wu, a0[u & 1]);
      a1[u & 1] = fmaf(v1[kc], wu, a1[u & 1]);
    }
  }
  r0 = a0[0] + a0[1];
  r1 = a1[0] + a1[1];
}

__device__ __forceinline__ float load_feat(const void* X, int bf16, int64_t idx);
__device__ __forceinline__ uint32_t feat_word(const void* X, int bf16, int64_t idx);
__device__ __forceinline__ float feat_value(uint32_t w, int bf16, int64_t idx);

// r[c] = sum_k v[k] W[k * ld + c] for c < n (W row-major [K, n], v in LDS; NV = 1 or 2 vectors against the same W) with 16-BYTE loads: a
// thread owns FOUR consecutive outputs and one of S interleaved K slices (S = 256 / (n / 4), capped), 8 rows in flight; the slices' partial
// sums meet in `part` ([S][NV][n] floats of LDS) and are added in slice order.  Measured (profiles/r06f): a 4-byte-per-lane load
// instruction costs these kernels ~24 cycles whatever it fetches, so the 200 x 300 Linear as one output per thread (200 - 400 scalar
// loads each) took 21 us of the 38 us forward kernel; a quarter of the instructions, each fetching 1 KB per wavefront, is what helps —
// not more loads in flight (64 instead of 16 made it slower).  Needs n % 4 == 0, ld % 4 == 0, W 16-byte aligned (mv4_ok); call from
// all 256 threads; results are in part[0 .. NV * n) afterwards (vector 0, then vector 1), valid after the trailing barrier.
__device__ __forceinline__ bool mv4_ok(const float* W, int64_t ld, int n) { return n % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(W) & 15u) == 0; }
__host__ __device__ inline int mv4_slices(int n) {
  const int q = n / 4;
  return q >= 256 ? 1 : (256 / q > 8 ? 8 : 256 / q);
}
template <int NV>
__device__ __forceinline__ void matvec4(const float* v0, const float* v1, const float* __restrict__ W, int64_t ld, int K, int n, float* part) {
  const int tid = threadIdx.x, Q = n >> 2, S = mv4_slices(n);
  for (int q0 = 0; q0 < Q; q0 += 256) {  // (n > 1024 would take several column passes; the callers cap n at 1024)
    const int q = q0 + (S > 1 ? tid % Q : tid), sl = S > 1 ? tid / Q : 0;
    const bool active = q < Q && sl < S;
    const int qc = min(q, Q - 1);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int k0 = sl; k0 < K; k0 += 8 * S) {
      float4 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(W + (int64_t)min(k0 + u * S, K - 1) * ld + 4 * qc);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kc = min(k0 + u * S, K - 1);
        float x0 = v0[kc];
        x0 = k0 + u * S < K ? x0 : 0.f;
        a0.x = fmaf(x0, w[u].x, a0.x); a0.y = fmaf(x0, w[u].y, a0.y); a0.z = fmaf(x0, w[u].z, a0.z); a0.w = fmaf(x0, w[u].w, a0.w);
        if (NV == 2) {
          float x1 = v1[kc];
          x1 = k0 + u * S < K ? x1 : 0.f;
          a1.x = fmaf(x1, w[u].x, a1.x); a1.y = fmaf(x1, w[u].y, a1.y); a1.z = fmaf(x1, w[u].z, a1.z); a1.w = fmaf(x1, w[u].w, a1.w);
        }
      }
    }
    if (active) {
      *reinterpret_cast<float4*>(part + ((int64_t)sl * NV) * n + 4 * q) = a0;
      if (NV == 2) *reinterpret_cast<float4*>(part + ((int64_t)sl * NV + 1) * n + 4 * q) = a1;
    }
  }
  __syncthreads();
  for (int e = tid; e < NV * n; e += 256) {
    float r = part[e];
    for (int sl = 1; sl < S; ++sl) r += part[(int64_t)sl * NV * n + e];  // slice order: deterministic
    part[e] = r;
  }
  __syncthreads();
}

// ---- Adam: counters and bias corrections (once per step, one thread) ----------------------------------------------------------------
struct AdamHyper {
  float lr, beta1, beta2, eps, wd;
};
constexpr int MS_MAX_PARAMS = 6;
struct StepCounters {
  float* step[MS_MAX_PARAMS];
  int n;
};
// coef[0] = lr / (1 - beta1^t), coef[1] = sqrt(1 - beta2^t) — evaluated in double and rounded once, as adam.hip does
__device__ void adam_tick(const StepCounters& sc, const AdamHyper& h, float* coef) {
  float t = 0.f;
  for (int j = 0; j < sc.n; ++j) {
    t = *sc.step[j] + 1.f;
    *sc.step[j] = t;
  }
  const double step = (double)t;
  coef[0] = h.lr / (float)(1.0 - pow((double)h.beta1, step));
  coef[1] = (float)sqrt(1.0 - pow((double)h.beta2, step));
}
// torch/optim/adam.py _single_tensor_adam, every operation rounded separately
__device__ __forceinline__ float adam_apply(float p, float g, float* m, float* v, const AdamHyper& h, float step_size, float bc2_sqrt) {
  const float one_minus_b1 = 1.f - h.beta1, one_minus_b2 = 1.f - h.beta2;
  if (h.wd != 0.f) g = g + h.wd * p;
  const float mi = *m + one_minus_b1 * (g - *m);
  const float vi = *v * h.beta2 + one_minus_b2 * g * g;
  *m = mi;
  *v = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + h.eps;
  return p - step_size * (mi / denom);
}

// =====================================================================================================================================
// graph-sc
// =====================================================================================================================================
struct GscArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  const float* X;  // node features [n_nodes, F], genes first
  int64_t ldx, n_nodes;
  int G, B, F, H, E, FP, mean;
  const float *W1, *b1, *W2, *b2;  // [F, H], [H], [E, H], [E]
  const float* w2t;                // [H, E] mirror of W2 (kept by ms_grad)
  int32_t* count;                  // [G] zero between steps
  float* coef;                     // [2]
  float* xdg;                      // [2, G, F] dropped gene rows (dropout > 0 only)
  float *ax2, *h2, *zd2, *demb, *dpre;  // [B, F], [B, H], [B, E], [B, E], [B, H]
  float* zd2t;                          // [E, B]: zd2 transposed (the decoder's row of z z^T as a 16-byte mat-vec)
  double* rowloss;                 // [B]
  int32_t* bad;
  float* ax_out;                   // phase 3: [2, B, F] — the aggregated (and D_in^-1/2-scaled) inputs of both forwards; the kernel stops there
  int dbg_fwd, dbg_dec;            // development: stop the forward / decoder kernel after phase n (DANCE_AMD_MINISTEP_DBG=f,d; 0 = run all)
};

// LDS_HIST (large batches): the count blocks are few and fat — every block walks B / nb_count seeds and counts into an LDS histogram,
// flushed with one global atomic per touched gene.  At 8192 seeds x 200 genes the one-atomic-per-entry form was 1.6 M adds onto 2000
// addresses: 0.2 ms, two thirds of the whole aggregation call (profiles/r06q); the histogram form is ~128 k.
template <bool LDS_HIST>
__global__ __launch_bounds__(256) void gsc_prepare_kernel(GscArgs a, const int64_t* __restrict__ seeds, Drop dx, StepCounters sc, AdamHyper hy, int nb_count) {
  extern __shared__ int hist[];  // [G] (LDS_HIST only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < nb_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && sc.n > 0) adam_tick(sc, hy, a.coef);
    if (LDS_HIST) {
      for (int g = threadIdx.x; g < a.G; g += 256) hist[g] = 0;
      __syncthreads();
    }
    if (LDS_HIST) {
      // Thousands of seeds over a few resident workgroups (their LDS histograms are what keeps 1.6 M atomics off the L2): a wave owns the
      // seeds w, w + W, w + 2 W, ... (W = all waves of the counting blocks).  Walking them one after the other was three DEPENDENT round
      // trips per seed — seed id -> row bounds -> column ids — 32 times in a row: 87 us per 8192-seed batch, 10.6 ms of graph-sc's 157 ms
      // large-batch epoch (profiles/r06z_graphsc_epoch_kernels_1M_b8192.md).  Now the lanes fetch the ids and bounds of up to 64 of
      // the wave's seeds at once (three round trips per WAVE), and a row's column ids are requested 256 at a time, two rows in flight.
      const int W = nb_count * 4, w0 = blockIdx.x * 4 + wave;
      for (int base = w0; base < a.B; base += 64 * W) {
        const int mine = base + lane * W;  // this lane's seed of the group
        int64_t v = mine < a.B ? seeds[mine] : -1;
        const bool alien = mine < a.B && (v < a.G || v >= a.n_nodes);
        if (alien) atomicOr(a.bad, 1);  // not a cell of this layout
        if (mine >= a.B || alien) v = -1;
        const int s_l = v >= 0 ? a.rowptr[v] : 0, t_l = v >= 0 ? a.rowptr[v + 1] : 0;
        const int n_rows = min(64, (a.B - base + W - 1) / W);
        auto row_of = [&](int k, int& s, int& t, int& vv) __attribute__((always_inline)) {
          s = __builtin_amdgcn_readlane(s_l, k);
          t = __builtin_amdgcn_readlane(t_l, k);
          vv = __builtin_amdgcn_readlane((int)v, k);
        };
        auto fetch = [&](int s, int t, int e0, int (&c)[4]) __attribute__((always_inline)) {  // 256 entries from e0 (clamped: always a valid address)
#pragma unroll
          for (int j = 0; j < 4; ++j) c[j] = t > s ? a.col[min(e0 + 64 * j + lane, t - 1)] : 0;
        };
        auto tally = [&](int s, int t, int vv, int e0, const int (&c)[4], int& n_self) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = e0 + 64 * j + lane;
            const bool in = e < t;
            if (in) {
              if (c[j] < a.G) atomicAdd(hist + c[j], 1);
              else if (c[j] != vv) atomicOr(a.bad, 1);  // a cell -> cell edge other than the self loop
            }
            n_self += __popcll(__ballot(in && c[j] >= a.G));
          }
        };
        int s0, t0, v0, cur[4];
        if (n_rows > 0) {
          row_of(0, s0, t0, v0);
          fetch(s0, t0, s0, cur);
        }
        for (int k = 0; k < n_rows; ++k) {
          int s1 = 0, t1 = 0, v1 = -1, nxt[4] = {0, 0, 0, 0};
          if (k + 1 < n_rows) {  // the next row's first 256 ids are in flight while this row is tallied
            row_of(k + 1, s1, t1, v1);
            fetch(s1, t1, s1, nxt);
          }
          int n_self = 0;
          if (v0 >= 0) {
            tally(s0, t0, v0, s0, cur, n_self);
            for (int e0 = s0 + 256; e0 < t0; e0 += 256) {  // rows beyond 256 entries: the rest on the spot
              int more[4];
              fetch(s0, t0, e0, more);
              tally(s0, t0, v0, e0, more, n_self);
            }
            if (lane == 0 && n_self != 1) atomicOr(a.bad, 2);  // the identity decoder target needs exactly one self loop per seed
          }
          s0 = s1, t0 = t1, v0 = v1;
#pragma unroll
          for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        }
      }
      __syncthreads();
      for (int g = threadIdx.x; g < a.G; g += 256) {
        const int h = hist[g];
        if (h) atomicAdd(a.count + g, h);
      }
      return;
    }
    for (int i = blockIdx.x * 4 + wave; i < a.B; i += a.B) {
      const int64_t v = seeds[i];
      if (v < a.G || v >= a.n_nodes) {  // not a cell of this layout
        if (lane == 0) atomicOr(a.bad, 1);
        continue;
      }
      const int s = a.rowptr[v], t = a.rowptr[v + 1];
      int n_self = 0;
      for (int e0 = s; e0 < t; e0 += 256) {  // 256 column ids requested together (one 64-id request per trip was a round trip each: 4 for a 200-gene cell)
        int c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = a.col[min(e0 + 64 * j + lane, t - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = e0 + 64 * j + lane < t;
          if (in) {
            if (c[j] < a.G) atomicAdd(a.count + c[j], 1);
            else if (c[j] != v) atomicOr(a.bad, 1);  // a cell -> cell edge other than the self loop
          }
          n_self += __popcll(__ballot(in && c[j] >= a.G));
        }
      }
      if (lane == 0 && n_self != 1) atomicOr(a.bad, 2);  // the identity decoder target needs exactly one self loop per seed
    }
    return;
  }
  if (dx.thr >= kKeepAll) return;
  // dropout of the gene rows, both forwards: one Philox call per 4 consecutive elements of the flat [G * F] index space
  const int64_t gf = (int64_t)a.G * a.F, nq = (gf + 3) >> 2;
  const int64_t q = ((int64_t)blockIdx.x - nb_count) * 256 + threadIdx.x;
  if (q >= 2 * nq) return;
  const int fwd = q >= nq;
  const int64_t quad = fwd ? q - nq : q;
  const uint4 r = drop_words(dx, SID_GENE + fwd, (uint64_t)quad);
  float xv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t e = min(quad * 4 + j, gf - 1), g = e / a.F, f = e - g * a.F;
    xv[j] = a.X[g * a.ldx + f];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (quad * 4 + j < gf) a.xdg[fwd * gf + quad * 4 + j] = xv[j] * drop_pick(dx, r, j);
}

// one workgroup per seed: BOTH forwards of the batch (graphsc.py:202 and :215) — they share every weight load
// AGG_ONLY (dh_graphsc_steps phase 3, large batches: thousands of seeds per launch): the kernel ends with the aggregated inputs, so it is
// built WITHOUT the dense phases' LDS and with 8 instead of 16 rows in flight — 186 registers and 30 KB of LDS held the whole-step form to
// two workgroups per CU, and a gather this latency-bound lives on resident wavefronts
template <bool AGG_ONLY>
__global__ __launch_bounds__(256) void gsc_forward_kernel(GscArgs a, const int64_t* __restrict__ seeds, Drop dx, Drop dd, float* __restrict__ z_out) {
  constexpr int EC = 1024;  // row entries staged per pass
  constexpr int U = AGG_ONLY ? 8 : 16;
  __shared__ float red[2][512], axs[2][128], hs[2][AGG_ONLY ? 1 : 1024], ews[EC];
  __shared__ __attribute__((aligned(16))) float part[AGG_ONLY ? 4 : 2048];  // matvec4: slices x 2 vectors x width <= 2 x 1024
  __shared__ int ecol[EC];
  const int i = blockIdx.x, tid = threadIdx.x;
  int64_t v = seeds[i];
  v = v < a.G ? a.G : v >= a.n_nodes ? a.n_nodes - 1 : v;  // (flagged by gsc_prepare; keep every address valid)
  const int s = a.rowptr[v], t = a.rowptr[v + 1], deg = t - s;
  const int FP = a.FP, ngrp = 256 / FP, fl = tid & (FP - 1), grp = tid / FP, F = a.F, G = a.G;
  const bool drop = dx.thr < kKeepAll;
  const float* __restrict__ xg0 = drop ? a.xdg : a.X;
  const float* __restrict__ xg1 = drop ? a.xdg + (int64_t)G * F : a.X;
  const int64_t ldg = drop ? F : a.ldx;
  const float* __restrict__ xself = a.X + v * a.ldx;
  const int f0 = min(fl, F - 1), f1 = min(fl + FP, F - 1);
  float sm[2][2] = {{1.f, 1.f}, {1.f, 1.f}};  // the seed row's own dropout draws [forward][feature slot]
  if (drop) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      sm[k][0] = drop_scale(dx, SID_SELF + k, (uint64_t)i * F + f0);
      sm[k][1] = drop_scale(dx, SID_SELF + k, (uint64_t)i * F + f1);
    }
  }
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  // rows that start on 8-byte boundaries are gathered as float2: one load per (row, forward) and thread instead of two 4-byte ones, and
  // twice the rows per wavefront instruction (the instruction count is what these latency-bound kernels pay for)
  const bool v2 = F % 2 == 0 && a.ldx % 2 == 0 && (reinterpret_cast<uintptr_t>(a.X) & 7u) == 0;
  const int P = F >> 1, PP = v2 ? min(pow2_at_least(P), 64) : FP, ngrp2 = 256 / PP, pl = tid & (PP - 1), grp2 = tid / PP, pc = min(pl, max(P, 1) - 1);
  float2 sm2[2] = {make_float2(1.f, 1.f), make_float2(1.f, 1.f)};
  if (v2 && drop) {
#pragma unroll
    for (int k = 0; k < 2; ++k) sm2[k] = make_float2(drop_scale(dx, SID_SELF + k, (uint64_t)i * F + 2 * pc), drop_scale(dx, SID_SELF + k, (uint64_t)i * F + 2 * pc + 1));
  }
  float2 acc2[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
  for (int c0 = s; c0 < t; c0 += EC) {
    const int n = min(EC, t - c0);
    // stage (source, weight x D_out^-1/2 of the source inside the block: graphsc.py:444-449 — a gene's count over the batch, 1 for the
    // seed's self loop) of up to EC entries: two round trips for the whole row instead of two per entry
    if (c0 > s) __syncthreads();
    for (int j = tid; j < n; j += 256) {
      const int c = a.col[c0 + j];
      const float w = a.val[c0 + j];
      const int cnt = a.count[min(c, G - 1)];
      ecol[j] = c;
      ews[j] = c >= G ? w : w * (1.f / sqrtf(fmaxf((float)cnt, 1.f)));
    }
    __syncthreads();
    if (v2) {
      for (int j0 = grp2; j0 < n; j0 += U * ngrp2) {
        float2 x[U][2];
        float w[U];
        bool self[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * ngrp2, jc = min(j, n - 1);
          const int c = ecol[jc];
          w[u] = j < n ? ews[jc] : 0.f;
          self[u] = c >= G;
          x[u][0] = *reinterpret_cast<const float2*>((self[u] ? xself : xg0 + (int64_t)c * ldg) + 2 * pc);
          if (drop) x[u][1] = *reinterpret_cast<const float2*>((self[u] ? xself : xg1 + (int64_t)c * ldg) + 2 * pc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (k == 1 && !drop) break;
            acc2[k].x = fmaf(w[u] * (self[u] ? sm2[k].x : 1.f), x[u][k].x, acc2[k].x);
            acc2[k].y = fmaf(w[u] * (self[u] ? sm2[k].y : 1.f), x[u][k].y, acc2[k].y);
          }
        }
      }
      continue;
    }
    // gather: 16 rows in flight per thread, every load from a clamped, valid address (what must not count gets a zero weight)
    for (int j0 = grp; j0 < n; j0 += U * ngrp) {
      float x[U][2][2], w[U];
      bool self[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u * ngrp, jc = min(j, n - 1);
        const int c = ecol[jc];
        w[u] = j < n ? ews[jc] : 0.f;
        self[u] = c >= G;
        const float* __restrict__ r0 = self[u] ? xself : xg0 + (int64_t)c * ldg;
        x[u][0][0] = r0[f0];
        x[u][0][1] = r0[f1];
        if (drop) {
          const float* __restrict__ r1 = self[u] ? xself : xg1 + (int64_t)c * ldg;
          x[u][1][0] = r1[f0];
          x[u][1][1] = r1[f1];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k == 1 && !drop) break;
          acc[k][0] = fmaf(w[u] * (self[u] ? sm[k][0] : 1.f), x[u][k][0], acc[k][0]);
          acc[k][1] = fmaf(w[u] * (self[u] ? sm[k][1] : 1.f), x[u][k][1], acc[k][1]);
        }
      }
    }
  }
  if (a.dbg_fwd == 1) return;
  if (!drop) {  // without dropout the two forwards aggregate the same thing
    acc[1][0] = acc[0][0];
    acc[1][1] = acc[0][1];
    acc2[1] = acc2[0];
  }
  const int ngrp_r = v2 ? ngrp2 : ngrp;
  if (v2) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (pl < P) {
        red[k][grp2 * F + 2 * pl] = acc2[k].x;
        red[k][grp2 * F + 2 * pl + 1] = acc2[k].y;
      }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (fl < F) red[k][grp * F + fl] = acc[k][0];
      if (fl + FP < F) red[k][grp * F + fl + FP] = acc[k][1];
    }
  }
  __syncthreads();
  // D_in^-1/2 of the destination (:467-471); fn.mean divides by the in-degree first (:465)
  const float dg = fmaxf((float)deg, 1.f);
  const float rs = (1.f / sqrtf(dg)) * (a.mean ? 1.f / dg : 1.f);
  for (int e = tid; e < 2 * F; e += 256) {
    const int k = e >= F, f = e - k * F;
    float sum = 0.f;
    for (int g2 = 0; g2 < ngrp_r; ++g2) sum += red[k][g2 * F + f];  // group order: deterministic
    sum *= rs;
    axs[k][f] = sum;
    if (k == 1) a.ax2[(int64_t)i * F + f] = sum;
    if (a.ax_out) a.ax_out[((int64_t)k * a.B + i) * F + f] = sum;
  }
  if (AGG_ONLY) return;  // (the large-batch loop runs the dense layers on the big-tile GEMMs)
  __syncthreads();
  const int H = a.H, E = a.E;
  const bool vb = mv4_ok(a.W1, H, H), vc = mv4_ok(a.w2t, E, E);
  if (vb) matvec4<2>(axs[0], axs[1], a.W1, H, F, H, part);
  for (int tt = tid; tt < H; tt += 256) {
    float p0, p1;
    if (vb) {
      p0 = part[tt];
      p1 = part[H + tt];
    } else {
      dot2_strided<16>(axs[0], axs[1], a.W1 + tt, H, F, p0, p1);
    }
    const float bb = a.b1[tt], h0 = fmaxf(p0 + bb, 0.f), h1 = fmaxf(p1 + bb, 0.f);
    hs[0][tt] = h0;
    hs[1][tt] = h1;
    a.h2[(int64_t)i * H + tt] = h1;
  }
  __syncthreads();
  if (a.dbg_fwd == 2) return;
  if (vc) matvec4<2>(hs[0], hs[1], a.w2t, E, H, E, part);
  for (int o = tid; o < E; o += 256) {
    float e0, e1;
    if (vc) {
      e0 = part[o];
      e1 = part[E + o];
    } else {
      dot2_strided<16>(hs[0], hs[1], a.w2t + o, E, H, e0, e1);
    }
    const float bb = a.b2[o];
    z_out[(int64_t)i * E + o] = e0 + bb;                                                             // graphsc.py:202-203
    const float zv = (e1 + bb) * drop_scale(dd, SID_DEC, (uint64_t)i * E + o);                    // :215, :409
    a.zd2[(int64_t)i * E + o] = zv;
    a.zd2t[(int64_t)o * a.B + i] = zv;
  }
}

template <typename Fn>
__device__ __forceinline__ void static_for2(Fn&& f) {
  f(std::integral_constant<int, 0>{});
  f(std::integral_constant<int, 1>{});
}

// ---- the aggregation of a LARGE batch on the matrix cores (phase 3, thousands of seeds) -----------------------------------------------
// AX = A_norm X_d as a dense product: 32 seed rows x G genes x F features per workgroup, the adjacency tile scattered into LDS chunk by
// chunk (128 genes) in MFMA operand order, the (dropped) gene rows streamed through LDS once per forward, exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32: every product w * x rounded once, fp32 accumulation — the gather kernel's arithmetic in another order).  At
// 10 % density the dense product spends 10x the useful flops and still wins by ~8x: the gather form fetches 201 rows of 200 bytes per
// seed and forward through L2 (660 MB per batch of 8192, 0.32 ms: profiles/r06k); here a workgroup reads the G x F table once per
// forward for 32 seeds (256 MB per batch) with 16-byte-class loads, and the 4000 MFMAs of a workgroup are 27 us of matrix-pipe time.
// The self loop (the seed's own row, its own dropout draw) is added on the vector ALUs in the epilogue.  Needs F <= 64 and at most
// AG_MAXE gene entries per seed row (the host checks the graph's maximum once); rows may list their entries in any order, each gene at
// most once (a CSR built from a matrix).
constexpr int AG_R = 32, AG_KC = 128, AG_MAXE = 256, AG_LDA = 33, AG_LDB = 65;

template <bool V2>
__global__ __launch_bounds__(256) void gsc_aggregate_mfma_kernel(GscArgs a, const int64_t* __restrict__ seeds, Drop dx) {
  extern __shared__ __attribute__((aligned(16))) char ag_smem[];
  float* As = reinterpret_cast<float*>(ag_smem);                  // [AG_KC][33]
  float* Bs = As + AG_KC * AG_LDA;                                // [AG_KC][65]
  int* eg = reinterpret_cast<int*>(Bs + AG_KC * AG_LDB);          // [32][AG_MAXE]
  float* ew = reinterpret_cast<float*>(eg + AG_R * AG_MAXE);      // [32][AG_MAXE]
  int* ecnt = reinterpret_cast<int*>(ew + AG_R * AG_MAXE);        // [32]
  float* wself = reinterpret_cast<float*>(ecnt + AG_R);           // [32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r0 = blockIdx.x * AG_R, F = a.F, G = a.G, B = a.B;
  const bool drop = dx.thr < kKeepAll;
  const int nf = drop ? 2 : 1;
  const int64_t ldg = drop ? F : a.ldx;
  // B tiles travel global -> registers -> LDS, one tile ahead of the products: the L2 round trip of tile (chunk, forward) + 1 runs
  // under the MFMAs of tile (chunk, forward).  A tile = AG_KC gene rows x F: NB pieces per thread (float2 when rows start on 8 bytes)
  constexpr int NB = V2 ? 16 : 32;
  const int P = V2 ? F >> 1 : F;  // pieces per gene row
  float2 breg[V2 ? NB : 1];
  float sreg[V2 ? 1 : NB];
  auto fetch_b = [&](int kchunk, int k) {
    const float* __restrict__ xg = drop ? a.xdg + (int64_t)k * G * F : a.X;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = min(tid + 256 * u, AG_KC * P - 1), kk = e / P, p = e - kk * P;
      const float* src = xg + (int64_t)min(kchunk + kk, G - 1) * ldg;
      if (V2) breg[u] = *reinterpret_cast<const float2*>(src + 2 * p);
      else sreg[u] = src[p];
    }
  };
  auto store_b = [&](int kchunk) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = tid + 256 * u;
      if (e < AG_KC * P) {
        const int kk = e / P, p = e - kk * P;
        const bool in = kchunk + kk < G;
        if (V2) {
          Bs[kk * AG_LDB + 2 * p] = in ? breg[u].x : 0.f;
          Bs[kk * AG_LDB + 2 * p + 1] = in ? breg[u].y : 0.f;
        } else {
          Bs[kk * AG_LDB + p] = in ? sreg[u] : 0.f;
        }
      }
    }
  };
  fetch_b(0, 0);  // in flight while the rows are staged
  if (tid < AG_R) wself[tid] = 0.f;
  // stage the rows' entries — (gene, weight x D_out^-1/2 of the gene inside the batch), the self loop's weight apart: a wave takes 8 rows,
  // 64 entries of each per round with all 16 (column, weight) loads in flight, then the 8 counter loads
  {
    int sb[8], tb[8], cntr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = min(r0 + wave * 8 + q, B - 1);
      int64_t v = seeds[i];
      v = v < G ? G : v >= a.n_nodes ? a.n_nodes - 1 : v;
      sb[q] = a.rowptr[v];
      tb[q] = a.rowptr[v + 1];
      cntr[q] = 0;
    }
    int longest = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) longest = max(longest, tb[q] - sb[q]);
    for (int e0 = 0; e0 < longest; e0 += 64) {
      int c[8], cn[8];
      float w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ec = min(sb[q] + e0 + lane, max(tb[q] - 1, sb[q]));
        c[q] = a.col[ec];
        w[q] = a.val[ec];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) cn[q] = a.count[min(c[q], G - 1)];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = wave * 8 + q;
        const bool in = sb[q] + e0 + lane < tb[q];
        const bool gene = in && c[q] < G;
        const unsigned long long m = __ballot(gene);
        if (gene) {
          const int pos = cntr[q] + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < AG_MAXE) {
            eg[r * AG_MAXE + pos] = c[q];
            ew[r * AG_MAXE + pos] = w[q] * (1.f / sqrtf(fmaxf((float)cn[q], 1.f)));
          }
        } else if (in) {
          wself[r] = w[q];  // (a cell row has exactly one self loop: checked by gsc_prepare)
        }
        cntr[q] = min(cntr[q] + (int)__popcll(m), AG_MAXE);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ecnt[wave * 8 + q] = cntr[q];
    }
  }
  __syncthreads();
  if (a.dbg_fwd == 3) return;
  const int i32 = lane & 31, h = lane >> 5, ntile = wave & 1, khalf = wave >> 1;
  f32x16 acc[2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const int srow = tid >> 3, sub = tid & 7;
  const int ne = ecnt[srow];
  for (int k0 = 0; k0 < G; k0 += AG_KC) {
    for (int e = tid; e < AG_KC * AG_LDA; e += 256) As[e] = 0.f;
    __syncthreads();
    if (a.dbg_fwd != 5)
    for (int j = sub; j < ne; j += 8) {
      const int g = eg[srow * AG_MAXE + j] - k0;
      if ((unsigned)g < (unsigned)AG_KC) As[g * AG_LDA + srow] = ew[srow * AG_MAXE + j];
    }
    // (the forward index is a compile-time constant in each copy of the body: acc[k] with a run-time k makes the compiler move 16
    // accumulator registers through selects around every MFMA — 800 cycles per step instead of 64, measured)
    static_for2([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (k < nf) {
        store_b(k0);
        __syncthreads();  // Bs (and, before the first forward, the scatter into As) published
        const bool last = k + 1 == nf;
        if (!last || k0 + AG_KC < G) fetch_b(last ? k0 + AG_KC : k0, last ? 0 : k + 1);  // the next tile, under this tile's products
        if (a.dbg_fwd != 4) {
#pragma unroll 4
          for (int s = khalf; s < AG_KC / 2; s += 2) {
            const float av = As[(2 * s + h) * AG_LDA + i32];
            const float bv = Bs[(2 * s + h) * AG_LDB + 32 * ntile + i32];
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
          }
        }
        __syncthreads();
      }
    });
  }
  // epilogue: the two K halves meet in LDS; + the self loop; x D_in^-1/2 (and 1 / in-degree for "mean")
  float* red = As;  // [khalf][fwd][32][64] = 8192 floats <= the As + Bs region (12544)
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((khalf * 2 + k) * AG_R + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * ntile + i32] = acc[k][r];
  __syncthreads();
  for (int e = tid; e < 2 * AG_R * F; e += 256) {
    const int k = e / (AG_R * F), rem = e - k * AG_R * F, r = rem / F, f = rem - r * F;
    const int i = r0 + r;
    if (i >= B) continue;
    int64_t v = seeds[i];
    v = v < G ? G : v >= a.n_nodes ? a.n_nodes - 1 : v;
    const int ks = drop ? k : 0;
    float sum = red[((0 * 2 + ks) * AG_R + r) * 64 + f] + red[((1 * 2 + ks) * AG_R + r) * 64 + f];
    const float sm = drop ? drop_scale(dx, SID_SELF + k, (uint64_t)i * F + f) : 1.f;
    sum = fmaf(wself[r] * sm, a.X[v * a.ldx + f], sum);
    const float dg = fmaxf((float)(a.rowptr[v + 1] - a.rowptr[v]), 1.f);
    sum *= (1.f / sqrtf(dg)) * (a.mean ? 1.f / dg : 1.f);
    a.ax_out[((int64_t)k * B + i) * F + f] = sum;
    if (k == 1) a.ax2[(int64_t)i * F + f] = sum;
  }
}

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// one workgroup per seed i: row i of the logits z z^T against the identity target, then everything of the backward that is local to row i
__global__ __launch_bounds__(256) void gsc_decoder_kernel(GscArgs a, Drop dd, float pos_weight, float cscale) {
  extern __shared__ float sm[];
  const int B = a.B, E = a.E, H = a.H, i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* zi = sm;          // [E]
  float* de = sm + E;      // [E]
  float* gj = sm + 2 * E;  // [B]
  __shared__ double wl[4];
  __shared__ __attribute__((aligned(16))) float part[1024];  // matvec4: slices x width <= 1024
  const float* __restrict__ Z = a.zd2;
  for (int kk = tid; kk < E; kk += 256) zi[kk] = Z[(int64_t)i * E + kk];
  __syncthreads();
  const bool v1 = B <= 1024 && mv4_ok(a.zd2t, B, B);
  if (v1) {  // x_ij = sum_k z_i[k] Z^T[k][j]: the 16-byte mat-vec over the transposed copy the forward kernel wrote
    matvec4<1>(zi, zi, a.zd2t, B, E, B, part);
    for (int j = tid; j < B; j += 256) gj[j] = part[j];
  } else {
    for (int j0 = wave; j0 < B; j0 += 32) {  // 8 rows per wavefront, 4 x 64 columns each: 32 loads in flight per lane
      float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < E; k0 += 256) {
        float zz[4], zr[8][4];
  #pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kk = k0 + lane + 64 * q, kc = min(kk, E - 1);
          zz[q] = zi[kc];
          zz[q] = kk < E ? zz[q] : 0.f;
  #pragma unroll
          for (int u = 0; u < 8; ++u) zr[u][q] = Z[(int64_t)min(j0 + 4 * u, B - 1) * E + kc];
        }
  #pragma unroll
        for (int q = 0; q < 4; ++q)
  #pragma unroll
          for (int u = 0; u < 8; ++u) d[u] = fmaf(zz[q], zr[u][q], d[u]);
      }
  #pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float x = wave_sum(d[u]);
        if (lane == 0 && j0 + 4 * u < B) gj[j0 + 4 * u] = x;  // the logit; its loss term and gradient below, one thread per column
      }
    }
  }
  __syncthreads();
  if (a.dbg_dec == 1) return;
  double loss = 0.0;
  for (int j = tid; j < B; j += 256) {
    const float x = gj[j];
    const float sg = 1.f / (1.f + expf(-x));
    if (j == i) {  // target 1, weighted: pos_weight * softplus(-x)
      loss += (double)(pos_weight * softplusf(-x));
      gj[j] = pos_weight * (sg - 1.f);
    } else {
      loss += (double)softplusf(x);
      gj[j] = sg;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off, 64);
  if (lane == 0) wl[wave] = loss;
  __syncthreads();
  if (tid == 0) a.rowloss[i] = ((wl[0] + wl[1]) + wl[2]) + wl[3];
  // d loss / d zd_i = 2 c sum_j g_ij zd_j (logits and target are symmetric), then the decoder dropout's own mask
  const bool v2 = mv4_ok(Z, E, E), v3 = mv4_ok(a.W2, H, H);
  if (v2) matvec4<1>(gj, gj, Z, E, B, E, part);
  for (int kk = tid; kk < E; kk += 256) {
    const float acc = v2 ? part[kk] : dot_strided<16>(gj, Z + kk, E, B);
    const float dv = 2.f * cscale * acc * drop_scale(dd, SID_DEC, (uint64_t)i * E + kk);
    de[kk] = dv;
    a.demb[(int64_t)i * E + kk] = dv;
  }
  __syncthreads();
  if (a.dbg_dec == 2) return;
  if (v3) matvec4<1>(de, de, a.W2, H, E, H, part);
  for (int tt = tid; tt < H; tt += 256) {
    const float dh = v3 ? part[tt] : dot_strided<16>(de, a.W2 + tt, H, E);
    a.dpre[(int64_t)i * H + tt] = a.h2[(int64_t)i * H + tt] > 0.f ? dh : 0.f;
  }
}

// =====================================================================================================================================
// weight gradients + Adam: G = A^T B in 32 x 32 tiles (A [K, M], B [K, N] row-major: the batch rows are the K dimension), exact fp32 MFMA
// =====================================================================================================================================
constexpr int MS_T = 32, MS_LD = 33;  // tile edge, LDS row stride; the K chunk per pass (KC) is a template parameter: 128 / 256 / 512 —
// the whole K extent in ONE round trip wherever it fits (a pass = two dependent global round trips of ~1 us + a barrier pair)

struct MsJob {
  const float* A;  // [K, M]; nullptr = a column of ones (M = 1: column sums, the bias gradients)
  int64_t lda;
  const void* Bm;  // [K, N] f32 or bf16
  int64_t ldb;
  const int64_t* b_rows;  // row k of B is row b_rows[k] of Bm (gathered feature rows), nullptr = k
  int b_bf16, b_drop;     // b_drop: B carries the SID_SDS dropout draw (element index k * N + n)
  int M, N, K;
  float *p, *m, *v;  // parameter [M, N] and its Adam moments
  float* pT;         // optional [N, M] mirror of the updated parameter
  float* g;          // optional gradient output [M, N] (data-parallel form: the update happens after the all-reduce)
  int tile0, tiles_n;
};
struct MsGradArgs {
  MsJob job[MS_MAX_PARAMS];
  int n_jobs, total_tiles, adam;
  AdamHyper hy;
  const float* coef;
  Drop bdrop;
  // extras (the block after the tiles): the step's loss and the counters to clear
  const double* rowloss_d;
  const float* rowloss_f;
  int n_rowloss;
  double loss_div;
  float loss_mul;
  float* loss_out;
  int32_t* zero_i32;
  int n_zero;
};

template <int MS_KC>
__global__ __launch_bounds__(256) void ms_grad_kernel(MsGradArgs a) {
  __shared__ float smem[2 * MS_KC * MS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= a.total_tiles) {
    // the step's loss: the rows' terms summed in double, in a fixed order (per-thread strided partial sums, then a tree over the 256
    // partials).  NOT one thread walking the rows: that is n dependent ~80 ns loads — 40 us at batch 500, the whole kernel's time.
    if (a.loss_out) {
      double* part = reinterpret_cast<double*>(smem);
      double sacc = 0.0;
      for (int i = tid; i < a.n_rowloss; i += 256) sacc += a.rowloss_d ? a.rowloss_d[i] : (double)a.rowloss_f[i];
      part[tid] = sacc;
      __syncthreads();
      for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
      }
      if (tid == 0) *a.loss_out = a.loss_mul * (float)(part[0] / a.loss_div);
    }
    for (int i = tid; i < a.n_zero; i += 256) a.zero_i32[i] = 0;
    return;
  }
  int ji = 0;
#pragma unroll
  for (int q = 1; q < MS_MAX_PARAMS; ++q)
    if (q < a.n_jobs && (int)blockIdx.x >= a.job[q].tile0) ji = q;
  const MsJob& jb = a.job[ji];
  const int tile = blockIdx.x - jb.tile0, m0 = (tile / jb.tiles_n) * MS_T, n0 = (tile % jb.tiles_n) * MS_T;
  const int M = jb.M, N = jb.N, K = jb.K;
  float* const As = smem;
  float* const Bs = smem + MS_KC * MS_LD;
  const int i32 = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool drop = jb.b_drop && a.bdrop.thr < kKeepAll;
  for (int k0 = 0; k0 < K; k0 += MS_KC) {
    float ra[MS_KC / 8], rb[MS_KC / 8];
    // (uniform branches sit OUTSIDE the unrolled batches: inside, every load shares one basic block and they all go out together)
    if (jb.A) {
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) {
        const int e = tid + 256 * q, r = e & 31, k = e >> 5;
        ra[q] = jb.A[(int64_t)min(k0 + k, K - 1) * jb.lda + min(m0 + r, M - 1)];
      }
    } else {
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) ra[q] = 1.f;
    }
    if (jb.b_rows) {
      int64_t brow[MS_KC / 8];
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) brow[q] = jb.b_rows[min(k0 + ((tid + 256 * q) >> 5), K - 1)];
      uint32_t wb[MS_KC / 8];
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) wb[q] = feat_word(jb.Bm, jb.b_bf16, brow[q] * jb.ldb + min(n0 + ((tid + 256 * q) & 31), N - 1));
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) rb[q] = feat_value(wb[q], jb.b_bf16, brow[q] * jb.ldb + min(n0 + ((tid + 256 * q) & 31), N - 1));
    } else {
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) {
        const int e = tid + 256 * q, r = e & 31, k = e >> 5;
        rb[q] = reinterpret_cast<const float*>(jb.Bm)[(int64_t)min(k0 + k, K - 1) * jb.ldb + min(n0 + r, N - 1)];
      }
    }
    if (drop) {
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) {
        const int e = tid + 256 * q, r = e & 31, k = e >> 5;
        rb[q] *= drop_scale(a.bdrop, SID_SDS, (uint64_t)min(k0 + k, K - 1) * N + min(n0 + r, N - 1));
      }
    }
    if (k0) __syncthreads();
#pragma unroll
    for (int q = 0; q < MS_KC / 8; ++q) {
      const int e = tid + 256 * q, r = e & 31, k = e >> 5;
      const bool kin = k0 + k < K;
      As[k * MS_LD + r] = (kin && m0 + r < M) ? ra[q] : 0.f;
      Bs[k * MS_LD + r] = (kin && n0 + r < N) ? rb[q] : 0.f;
    }
    __syncthreads();
    const int kc = min(MS_KC, K - k0);
    for (int s = wave; 2 * s < kc; s += 4) {
      const float av = As[(2 * s + h) * MS_LD + i32];
      const float bv = Bs[(2 * s + h) * MS_LD + i32];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  float* const redt = smem + wave * (MS_T * MS_LD);
#pragma unroll
  for (int r = 0; r < 16; ++r) redt[((r & 3) + 8 * (r >> 2) + 4 * h) * MS_LD + i32] = acc[r];
  __syncthreads();
  const float step_size = a.adam ? a.coef[0] : 0.f, bc2 = a.adam ? a.coef[1] : 1.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, row = e >> 5, cc = e & 31;
    float g = smem[row * MS_LD + cc];
#pragma unroll
    for (int w = 1; w < 4; ++w) g += smem[w * (MS_T * MS_LD) + row * MS_LD + cc];  // wave order: deterministic
    if (m0 + row < M && n0 + cc < N) {
      const int64_t idx = (int64_t)(m0 + row) * N + n0 + cc;
      if (jb.g) jb.g[idx] = g;
      if (a.adam) {
        const float pn = adam_apply(jb.p[idx], g, jb.m + idx, jb.v + idx, a.hy, step_size, bc2);
        jb.p[idx] = pn;
        if (jb.pT) jb.pT[(int64_t)(n0 + cc) * M + m0 + row] = pn;
      }
    }
  }
}

// the update alone, over the same job table (data-parallel form: after the gradient all-reduce)
__global__ __launch_bounds__(256) void ms_adam_kernel(MsGradArgs a) {
  const MsJob& jb = a.job[blockIdx.y];
  const int64_t n = (int64_t)jb.M * jb.N, idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float pn = adam_apply(jb.p[idx], jb.g[idx], jb.m + idx, jb.v + idx, a.hy, a.coef[0], a.coef[1]);
  jb.p[idx] = pn;
  if (jb.pT) jb.pT[(idx % jb.N) * jb.M + idx / jb.N] = pn;
}

__global__ __launch_bounds__(256) void ms_transpose_kernel(int M, int N, const float* __restrict__ p, float* __restrict__ pT) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (int64_t)M * N) pT[(idx % N) * M + idx / N] = p[idx];
}

int place_tiles(MsGradArgs& g) {
  int t = 0;
  for (int j = 0; j < g.n_jobs; ++j) {
    g.job[j].tile0 = t;
    g.job[j].tiles_n = (g.job[j].N + MS_T - 1) / MS_T;
    t += ((g.job[j].M + MS_T - 1) / MS_T) * g.job[j].tiles_n;
  }
  g.total_tiles = t;
  return t;
}

void launch_grad(const MsGradArgs& ga, int K, hipStream_t st) {
  const dim3 grid((unsigned)(ga.total_tiles + 1));
  if (K <= 128) hipLaunchKernelGGL(ms_grad_kernel<128>, grid, dim3(256), 0, st, ga);
  else hipLaunchKernelGGL(ms_grad_kernel<256>, grid, dim3(256), 0, st, ga);  // (512 per pass: 128 values in flight per thread spill)
}

Drop make_drop(float p, uint64_t seed, uint64_t step) {
  Drop d;
  d.seed_lo = (uint32_t)seed;
  d.seed_hi = (uint32_t)(seed >> 32);
  d.step_lo = (uint32_t)step;
  d.step_hi = (uint32_t)(step >> 32);
  if (p <= 0.f) {
    d.thr = kKeepAll;
    d.scale = 1.f;
  } else {
    const double keep = 1.0 - (double)p;
    d.thr = (uint32_t)(keep * 16777216.0);
    d.scale = (float)(1.0 / keep);
  }
  return d;
}

// =====================================================================================================================================
// scDeepSort
// =====================================================================================================================================
struct SdsArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  const void* X;  // node features [n_nodes, D], f32 or bf16
  int64_t ldx, n_nodes;
  int x_bf16, G, B, D, H, C;
  const int32_t* cell_id;  // [n_nodes]: gene index for genes, -1 for cells (the graph's ndata["cell_id"])
  const int64_t* labels;   // [n_nodes]
  const float* alpha;      // [G + 2]
  const float *W1, *b1, *W2, *b2;  // [H, D], [H], [C, H], [C]
  float* coef;
  float *h1, *dh1, *dlog, *rowloss;  // [B, H], [B, H], [B, C], [B]
  float* neigh;                      // [B, D] or nullptr
  int32_t* bad;
};

// one feature value, fp32 or bf16 storage, WITHOUT a branch on the type: a branch — even a uniform one — between the loads of an unrolled batch
// puts every load into its own basic block behind an s_waitcnt (scripts/isa_load_audit.py found runs of 64 - 128 dependent round trips in the
// first form of these kernels).  bf16: the aligned 32-bit word that holds the element, then a shift / mask.
__device__ __forceinline__ float load_feat(const void* X, int bf16, int64_t idx) {
  const uint32_t w = reinterpret_cast<const uint32_t*>(X)[bf16 ? (idx >> 1) : idx];
  return __uint_as_float(bf16 ? ((idx & 1) ? (w & 0xFFFF0000u) : (w << 16)) : w);
}
// the two halves of load_feat for batches: all raw words first (nothing touches a loaded value until every load of the batch is out —
// the compiler does not hoist a load above the select that consumes the previous one), conversion afterwards
__device__ __forceinline__ uint32_t feat_word(const void* X, int bf16, int64_t idx) { return reinterpret_cast<const uint32_t*>(X)[bf16 ? (idx >> 1) : idx]; }
__device__ __forceinline__ float feat_value(uint32_t w, int bf16, int64_t idx) {
  return __uint_as_float(bf16 ? ((idx & 1) ? (w & 0xFFFF0000u) : (w << 16)) : w);
}

// neigh[i] = mean over the in-edges e of seed i of alpha[idx(e)] w_e h[src(e)]  (gnn.py:62-90): alpha index = the gene's id for a
// gene -> cell edge, G + 1 for a cell's self loop (G for gene - gene, the destination gene's id for cell -> gene: not reachable from a seed cell)
__global__ __launch_bounds__(256) void sds_neigh_kernel(SdsArgs a, const int64_t* __restrict__ seeds) {
  extern __shared__ float sm[];  // [ngrp][D]
  const int i = blockIdx.x, tid = threadIdx.x, D = a.D;
  int64_t v = seeds[i];
  v = v < 0 ? 0 : v >= a.n_nodes ? a.n_nodes - 1 : v;
  const int s = a.rowptr[v], t = a.rowptr[v + 1];
  const int DP = min(pow2_at_least(D), 256);
  const int ngrp = 256 / DP, fl = tid & (DP - 1), grp = tid / DP;
  const int did = a.cell_id[v];
  constexpr int NF = 4;  // features per thread (D <= 1024)
  float acc[NF];
#pragma unroll
  for (int q = 0; q < NF; ++q) acc[q] = 0.f;
  for (int e0 = s + grp; e0 < t; e0 += 2 * ngrp) {
    int c[2];
    float w[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = e0 + u * ngrp, ec = min(e, t - 1);
      c[u] = a.col[ec];
      w[u] = e < t ? a.val[ec] : 0.f;
    }
    int sid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) sid[u] = a.cell_id[c[u]];
    float x[2][NF];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < NF; ++q) x[u][q] = load_feat(a.X, a.x_bf16, (int64_t)c[u] * a.ldx + min(fl + q * DP, D - 1));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ai = (sid[u] >= 0 && did < 0) ? sid[u] : (did >= 0 && sid[u] < 0) ? did : (did >= 0 && sid[u] >= 0) ? a.G : a.G + 1;
      const float ws = w[u] * a.alpha[ai];
#pragma unroll
      for (int q = 0; q < NF; ++q) acc[q] = fmaf(ws, x[u][q], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < NF; ++q)
    if (fl + q * DP < D) sm[grp * D + fl + q * DP] = acc[q];
  __syncthreads();
  const float inv = 1.f / fmaxf((float)(t - s), 1.f);
  for (int f = tid; f < D; f += 256) {
    float sum = 0.f;
    for (int g2 = 0; g2 < ngrp; ++g2) sum += sm[g2 * D + f];
    a.neigh[(int64_t)i * D + f] = sum * inv;
  }
}

// the same aggregation for 16-byte-aligned rows (D % 4 == 0): one wavefront per entry, four entries per wavefront in flight, lanes across
// the row in 4-feature vectors (NQ vectors per lane).  The scalar kernel above walks a row's ~200 entries two at a time behind
// col -> cell_id -> alpha -> feature round trips: 80 us at batch 500 x D 400; this one 16 entries per round.
template <int NQ, bool BF16>
__global__ __launch_bounds__(256) void sds_neigh_vec_kernel(SdsArgs a, const int64_t* __restrict__ seeds) {
  extern __shared__ float sm[];  // [4][D]
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, D = a.D, D4 = D >> 2;
  int64_t v = seeds[i];
  v = v < 0 ? 0 : v >= a.n_nodes ? a.n_nodes - 1 : v;
  const int s = a.rowptr[v], t = a.rowptr[v + 1];
  const int did = a.cell_id[v];
  float4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int U = 4;
  for (int e0 = s + wave; e0 < t; e0 += 4 * U) {
    int c[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 4 * u, ec = min(e, t - 1);
      c[u] = a.col[ec];
      w[u] = e < t ? a.val[ec] : 0.f;
    }
    int sid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sid[u] = a.cell_id[c[u]];
    float4 x[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int64_t at = (int64_t)c[u] * a.ldx + 4 * min(lane + 64 * q, D4 - 1);
        if (BF16) {
          const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.X) + at);
          x[u][q] = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
        } else {
          x[u][q] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.X) + at);
        }
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ai = (sid[u] >= 0 && did < 0) ? sid[u] : (did >= 0 && sid[u] < 0) ? did : (did >= 0 && sid[u] >= 0) ? a.G : a.G + 1;
      const float ws = w[u] * a.alpha[ai];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        acc[q].x = fmaf(ws, x[u][q].x, acc[q].x);
        acc[q].y = fmaf(ws, x[u][q].y, acc[q].y);
        acc[q].z = fmaf(ws, x[u][q].z, acc[q].z);
        acc[q].w = fmaf(ws, x[u][q].w, acc[q].w);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + 64 * q < D4) *reinterpret_cast<float4*>(sm + wave * D + 4 * (lane + 64 * q)) = acc[q];
  __syncthreads();
  const float inv = 1.f / fmaxf((float)(t - s), 1.f);
  for (int f = tid; f < D; f += 256) a.neigh[(int64_t)i * D + f] = (((sm[f] + sm[D + f]) + sm[2 * D + f]) + sm[3 * D + f]) * inv;
}

// h1 = relu(dropout(X[seeds]) W1^T + b1): 32 x 32 tiles, the feature rows gathered on the way into LDS
template <int MS_KC>
__global__ __launch_bounds__(256) void sds_hidden_kernel(SdsArgs a, const int64_t* __restrict__ seeds, Drop dz, StepCounters sc, AdamHyper hy) {
  __shared__ float smem[2 * MS_KC * MS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) adam_tick(sc, hy, a.coef);
  const int m0 = blockIdx.y * MS_T, n0 = blockIdx.x * MS_T, M = a.B, N = a.H, K = a.D;
  float* const As = smem;
  float* const Bs = smem + MS_KC * MS_LD;
  const int i32 = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool drop = dz.thr < kKeepAll;
  const float bias_c = a.b1[min(n0 + (tid & 31), N - 1)];  // this thread's output column (the same for its four rows): requested first, used last
  // the tile's 32 seeds: ONE load per lane through LDS (read per element they become 32 dependent scalar round trips: the row index is
  // uniform over the workgroup and the compiler moves it to an SGPR behind a wait)
  __shared__ int srow[MS_T];
  if (tid < MS_T) {
    const int64_t v = seeds[min(m0 + tid, M - 1)];
    srow[tid] = (int)(v < 0 ? 0 : v >= a.n_nodes ? a.n_nodes - 1 : v);
  }
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += MS_KC) {
    float ra[MS_KC / 8], rb[MS_KC / 8];
    uint32_t wa[MS_KC / 8];
#pragma unroll
    for (int q = 0; q < MS_KC / 8; ++q) {  // both operands K-contiguous: consecutive lanes walk k
      const int e = tid + 256 * q, r = e / MS_KC, k = e % MS_KC;
      const int kk = min(k0 + k, K - 1);
      wa[q] = feat_word(a.X, a.x_bf16, (int64_t)srow[r] * a.ldx + kk);
      rb[q] = a.W1[(int64_t)min(n0 + r, N - 1) * K + kk];
    }
#pragma unroll
    for (int q = 0; q < MS_KC / 8; ++q) {
      const int e = tid + 256 * q, r = e / MS_KC, k = e % MS_KC;
      ra[q] = feat_value(wa[q], a.x_bf16, (int64_t)srow[r] * a.ldx + min(k0 + k, K - 1));
    }
    if (drop) {
#pragma unroll
      for (int q = 0; q < MS_KC / 8; ++q) {
        const int e = tid + 256 * q, r = e / MS_KC, k = e % MS_KC;
        ra[q] *= drop_scale(dz, SID_SDS, (uint64_t)min(m0 + r, M - 1) * K + min(k0 + k, K - 1));
      }
    }
    if (k0) __syncthreads();
#pragma unroll
    for (int q = 0; q < MS_KC / 8; ++q) {
      const int e = tid + 256 * q, r = e / MS_KC, k = e % MS_KC;
      const bool kin = k0 + k < K;
      As[k * MS_LD + r] = (kin && m0 + r < M) ? ra[q] : 0.f;
      Bs[k * MS_LD + r] = (kin && n0 + r < N) ? rb[q] : 0.f;
    }
    __syncthreads();
    const int kc = min(MS_KC, K - k0);
    for (int s = wave; 2 * s < kc; s += 4) {
      const float av = As[(2 * s + h) * MS_LD + i32];
      const float bv = Bs[(2 * s + h) * MS_LD + i32];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  float* const redt = smem + wave * (MS_T * MS_LD);
#pragma unroll
  for (int r = 0; r < 16; ++r) redt[((r & 3) + 8 * (r >> 2) + 4 * h) * MS_LD + i32] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, row = e >> 5, cc = e & 31;
    float vv = smem[row * MS_LD + cc];
#pragma unroll
    for (int w = 1; w < 4; ++w) vv += smem[w * (MS_T * MS_LD) + row * MS_LD + cc];
    if (m0 + row < M && n0 + cc < N) a.h1[(int64_t)(m0 + row) * N + n0 + cc] = fmaxf(vv + bias_c, 0.f);
  }
}

// one wavefront per seed: logits = h1 W2^T + b2, loss_i = logsumexp - logit[label], dlog = softmax - onehot (CrossEntropyLoss(reduction="sum"),
// scdeepsort.py:185), dh1 = (dlog W2) masked by the ReLU.  Lane = (class c, K quarter q): CP classes padded to a power of two, 64 / CP quarters.
__global__ __launch_bounds__(256) void sds_loss_kernel(SdsArgs a, const int64_t* __restrict__ seeds) {
  extern __shared__ float sm[];  // W2 [C][H + 1], then per wave: h [H], dl [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, H = a.H, C = a.C, HL = H + 1;
  float* w2s = sm;
  float* hw = sm + C * HL + wave * (H + 64);
  float* dlw = hw + H;
  const int i = blockIdx.x * 4 + wave;
  const bool live = i < a.B;
  const int ii = live ? i : a.B - 1;
  // the seed -> label chain (two dependent round trips) starts before the staging loops instead of after them
  int64_t v = seeds[ii];
  v = v < 0 ? 0 : v >= a.n_nodes ? a.n_nodes - 1 : v;
  const int64_t lab = a.labels[v];
  // W2 into LDS, eight elements per thread in flight (one per trip of a plain loop: 13 dependent round trips for 16 x 200 weights —
  // most of this kernel's 11 us; found by listing loops whose only load sits in front of a wait)
  {
    const int n_w = C * H;
    int e = tid;
    for (; e + 7 * 256 < n_w; e += 8 * 256) {
      float w8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w8[u] = a.W2[e + 256 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) w2s[((e + 256 * u) / H) * HL + (e + 256 * u) % H] = w8[u];
    }
    float w8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w8[u] = a.W2[min(e + 256 * u, n_w - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e + 256 * u < n_w) w2s[((e + 256 * u) / H) * HL + (e + 256 * u) % H] = w8[u];
  }
  {
    float h4[4];  // H <= 256: the row in one batch
    int kk = lane;
    for (; kk + 3 * 64 < H; kk += 4 * 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) h4[u] = a.h1[(int64_t)ii * H + kk + 64 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u) hw[kk + 64 * u] = h4[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) h4[u] = a.h1[(int64_t)ii * H + min(kk + 64 * u, H - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (kk + 64 * u < H) hw[kk + 64 * u] = h4[u];
  }
  __syncthreads();
  const int CP = pow2_at_least(C), Q = 64 / CP, c = lane & (CP - 1), q = lane / CP;
  const int hq = (H + Q - 1) / Q, kb = q * hq, ke = min(H, kb + hq);
  float part = 0.f;
  const int cc = min(c, C - 1);
  for (int kk = kb; kk < ke; ++kk) part = fmaf(hw[kk], w2s[cc * HL + kk], part);
  for (int off = CP; off < 64; off <<= 1) part += __shfl_xor(part, off, 64);  // over the quarters
  float logit = c < C ? part + a.b2[cc] : -INFINITY;
  float mx = logit;
  for (int off = 1; off < CP; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  const float ex = c < C ? expf(logit - mx) : 0.f;
  float se = ex;
  for (int off = 1; off < CP; off <<= 1) se += __shfl_xor(se, off, 64);
  if (live && lane == 0 && (lab < 0 || lab >= C)) atomicOr(a.bad, 4);
  const float dl = c < C ? ex / se - (c == lab ? 1.f : 0.f) : 0.f;
  float lt = (c == lab) ? logit : 0.f;  // the label's logit, broadcast over the wavefront's class lanes
  for (int off = 1; off < CP; off <<= 1) lt += __shfl_xor(lt, off, 64);
  if (lane < CP) dlw[lane] = dl;
  if (live && lane == 0) a.rowloss[i] = (logf(se) + mx) - lt;
  if (live && q == 0 && c < C) a.dlog[(int64_t)i * C + c] = dl;
  __syncthreads();
  for (int kk = lane; kk < H; kk += 64) {
    float acc = 0.f;
    for (int c2 = 0; c2 < C; ++c2) acc = fmaf(dlw[c2], w2s[c2 * HL + kk], acc);
    if (live) a.dh1[(int64_t)i * H + kk] = hw[kk] > 0.f ? acc : 0.f;
  }
}

}  // namespace

// =====================================================================================================================================
// host side
// =====================================================================================================================================
namespace {

__global__ __launch_bounds__(256) void ms_dropout_mask_kernel(int64_t n, Drop d, uint32_t sid, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) out[e] = drop_scale(d, sid, (uint64_t)e);
}

size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }

struct GscLayout {
  size_t count, coef, xdg, ax2, h2, zd2, zd2t, demb, dpre, rowloss, w2t, total;
};
GscLayout gsc_layout(int64_t G, int64_t B, int64_t F, int64_t H, int64_t E) {
  GscLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += a256(bytes);
    return at;
  };
  l.count = take((size_t)G * 4);
  l.coef = take(16);
  l.xdg = take((size_t)2 * G * F * 4);
  l.ax2 = take((size_t)B * F * 4);
  l.h2 = take((size_t)B * H * 4);
  l.zd2 = take((size_t)B * E * 4);
  l.zd2t = take((size_t)B * E * 4);
  l.demb = take((size_t)B * E * 4);
  l.dpre = take((size_t)B * H * 4);
  l.rowloss = take((size_t)B * 8);
  l.w2t = take((size_t)H * E * 4);
  l.total = o;
  return l;
}

bool adam_ok(const dh_adam_state_t& s) { return s.param && s.exp_avg && s.exp_avg_sq && s.step; }

void set_job(MsJob& j, const float* A, int64_t lda, const void* Bm, int64_t ldb, int M, int N, int K, const dh_adam_state_t& st, float* pT, float* g) {
  j = MsJob{};
  j.A = A;
  j.lda = lda;
  j.Bm = Bm;
  j.ldb = ldb;
  j.M = M;
  j.N = N;
  j.K = K;
  j.p = st.param;
  j.m = st.exp_avg;
  j.v = st.exp_avg_sq;
  j.pT = pT;
  j.g = g;
}

}  // namespace

extern "C" int dh_ministep_dropout_mask_f32(int64_t n, float p, uint64_t seed, uint64_t step, int32_t sid, float* out, dh_stream_t stream) {
  const char* me = "dh_ministep_dropout_mask_f32";
  if (n < 0 || p < 0.f || p >= 1.f || sid < 0 || sid > 255) return dh::fail(DH_ERR_INVALID, "%s: bad argument", me);
  if (n == 0) return DH_OK;
  if (!out) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  hipLaunchKernelGGL(ms_dropout_mask_kernel, dim3((unsigned)dh::ceil_div(n, 256)), dim3(256), 0, dh::as_stream(stream), n, make_drop(p, seed, step), (uint32_t)sid,
                     out);
  return dh::check_launch(me);
}

extern "C" int dh_graphsc_step_supported(int64_t batch, int64_t in_feats, int64_t hidden, int64_t emb) {
  return (batch >= 2 && batch <= 65536 && in_feats >= 1 && in_feats <= 128 && hidden >= 1 && hidden <= 1024 && emb >= 1 && emb <= 1024) ? 1 : 0;
}

extern "C" size_t dh_graphsc_step_workspace_bytes(int64_t n_genes, int64_t batch, int64_t in_feats, int64_t hidden, int64_t emb) {
  if (n_genes < 1 || !dh_graphsc_step_supported(batch, in_feats, hidden, emb)) return 0;
  return gsc_layout(n_genes, batch, in_feats, hidden, emb).total;
}

extern "C" int dh_graphsc_steps(const dh_graphsc_step_t* c, int64_t first_step, int64_t n_steps, dh_stream_t stream) {
  const char* me = "dh_graphsc_steps";
  if (!c) return dh::fail(DH_ERR_INVALID, "%s: null configuration", me);
  if (first_step < 0 || n_steps < 0) return dh::fail(DH_ERR_INVALID, "%s: negative step range", me);
  if (n_steps == 0) return DH_OK;
  if (!dh_graphsc_step_supported(c->batch, c->in_feats, c->hidden, c->emb))
    return dh::fail(DH_ERR_INVALID, "%s: batch %lld, %lld -> %lld -> %lld outside dh_graphsc_step_supported", me, (long long)c->batch, (long long)c->in_feats,
                    (long long)c->hidden, (long long)c->emb);
  if (c->n_genes < 1 || c->n_nodes <= c->n_genes || c->n_nodes >= ((int64_t)1 << 31)) return dh::fail(DH_ERR_INVALID, "%s: bad node counts", me);
  if (!c->rowptr || !c->col || !c->val || !c->features || !c->seeds || !c->bad || !c->workspace) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (c->ld_features < c->in_feats) return dh::fail(DH_ERR_INVALID, "%s: ld_features < in_feats", me);
  if (!adam_ok(c->w1) || !adam_ok(c->b1) || !adam_ok(c->w2) || !adam_ok(c->b2)) return dh::fail(DH_ERR_INVALID, "%s: incomplete Adam state", me);
  if (c->dropout < 0.f || c->dropout >= 1.f || c->decoder_dropout < 0.f || c->decoder_dropout >= 1.f) return dh::fail(DH_ERR_INVALID, "%s: dropout outside [0, 1)", me);
  if (c->phase < 0 || c->phase > 3 || ((c->phase == 1 || c->phase == 2) && (n_steps != 1 || !c->grads)))
    return dh::fail(DH_ERR_INVALID, "%s: phases 1 / 2 take one step and a gradient buffer", me);
  if (c->phase == 3 && (n_steps != 1 || !c->ax_out)) return dh::fail(DH_ERR_INVALID, "%s: phase 3 (aggregate only) takes one step and ax_out", me);
  if (c->phase < 2 && (!c->z_out || !c->loss_out)) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  if (c->phase < 2 && (2 * c->emb + c->batch) * 4 > 48 * 1024) return dh::fail(DH_ERR_INVALID, "%s: batch %lld too large for the one-workgroup-per-seed decoder (use phase 3 + the all-pairs kernels)", me, (long long)c->batch);
  const int B = (int)c->batch, F = (int)c->in_feats, H = (int)c->hidden, E = (int)c->emb, G = (int)c->n_genes;
  const GscLayout l = gsc_layout(G, B, F, H, E);
  if (c->workspace_bytes < l.total) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, c->workspace_bytes, l.total);
  hipStream_t st = dh::as_stream(stream);
  char* ws = static_cast<char*>(c->workspace);
  GscArgs a{};
  a.rowptr = c->rowptr; a.col = c->col; a.val = c->val; a.X = c->features; a.ldx = c->ld_features; a.n_nodes = c->n_nodes;
  a.G = G; a.B = B; a.F = F; a.H = H; a.E = E; a.FP = std::min(pow2_at_least(F), 64); a.mean = c->agg_mean ? 1 : 0;
  a.W1 = c->w1.param; a.b1 = c->b1.param; a.W2 = c->w2.param; a.b2 = c->b2.param;
  float* w2t = reinterpret_cast<float*>(ws + l.w2t);
  a.w2t = w2t;
  a.count = reinterpret_cast<int32_t*>(ws + l.count);
  a.coef = reinterpret_cast<float*>(ws + l.coef);
  a.xdg = reinterpret_cast<float*>(ws + l.xdg);
  a.ax2 = reinterpret_cast<float*>(ws + l.ax2);
  a.h2 = reinterpret_cast<float*>(ws + l.h2);
  a.zd2 = reinterpret_cast<float*>(ws + l.zd2);
  a.zd2t = reinterpret_cast<float*>(ws + l.zd2t);
  a.demb = reinterpret_cast<float*>(ws + l.demb);
  a.dpre = reinterpret_cast<float*>(ws + l.dpre);
  a.rowloss = reinterpret_cast<double*>(ws + l.rowloss);
  a.bad = c->bad;
  if (const char* dbg = getenv("DANCE_AMD_MINISTEP_DBG")) sscanf(dbg, "%d,%d", &a.dbg_fwd, &a.dbg_dec);
  const AdamHyper hy{c->lr, c->beta1, c->beta2, c->eps, c->weight_decay};
  StepCounters sc{};
  sc.step[0] = c->w1.step; sc.step[1] = c->b1.step; sc.step[2] = c->w2.step; sc.step[3] = c->b2.step;
  sc.n = 4;
  // the job table of the gradient kernel (constant over the steps)
  MsGradArgs ga{};
  float* g = c->phase ? c->grads : nullptr;
  const int64_t o_b1 = (int64_t)F * H, o_w2 = o_b1 + H, o_b2 = o_w2 + (int64_t)E * H;
  set_job(ga.job[0], a.ax2, F, a.dpre, H, F, H, B, c->w1, nullptr, g);                       // dW1 = AX^T dPre
  set_job(ga.job[1], nullptr, 0, a.dpre, H, 1, H, B, c->b1, nullptr, g ? g + o_b1 : nullptr); // db1
  set_job(ga.job[2], a.demb, E, a.h2, H, E, H, B, c->w2, w2t, g ? g + o_w2 : nullptr);        // dW2 = dEmb^T h
  set_job(ga.job[3], nullptr, 0, a.demb, E, 1, E, B, c->b2, nullptr, g ? g + o_b2 : nullptr); // db2
  ga.n_jobs = 4;
  place_tiles(ga);
  ga.adam = c->phase == 0;
  ga.hy = hy;
  ga.coef = a.coef;
  ga.bdrop = make_drop(0.f, 0, 0);
  ga.rowloss_d = a.rowloss;
  ga.n_rowloss = B;
  const double b = (double)B;
  const double pos_weight = (b * b - b) / b, norm = b * b / ((b * b - b) * 2.0);  // graphsc.py:210-214 with adj = I
  ga.loss_div = b * b;
  ga.loss_mul = (float)norm;
  ga.zero_i32 = a.count;
  ga.n_zero = G;
  if (c->phase == 2) {
    int64_t longest = 0;
    for (int j = 0; j < ga.n_jobs; ++j) longest = std::max<int64_t>(longest, (int64_t)ga.job[j].M * ga.job[j].N);
    hipLaunchKernelGGL(ms_adam_kernel, dim3((unsigned)dh::ceil_div(longest, 256), (unsigned)ga.n_jobs), dim3(256), 0, st, ga);
    return dh::check_launch(me);
  }
  if (c->phase == 3) {  // aggregate only: counters, both forwards' AX (no Adam tick, no dense layer)
    if (dh::zero_async(a.count, (size_t)G * 4, st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: clearing the counters failed", me);
    const bool drop3 = c->dropout > 0.f;
    const int nbc = (B + 3) / 4, nbd = drop3 ? (int)dh::ceil_div(2 * (((int64_t)G * F + 3) / 4), 256) : 0;
    const Drop dx3 = make_drop(c->dropout, c->seed, c->step0 + (uint64_t)first_step);
    StepCounters none{};
    a.ax_out = c->ax_out;
    const int64_t* seeds3 = c->seeds + first_step * B;
    if (G <= 12288 && B >= 1024) {
      const int nbh = 64;
      hipLaunchKernelGGL(gsc_prepare_kernel<true>, dim3((unsigned)(nbh + nbd)), dim3(256), (size_t)G * 4, st, a, seeds3, dx3, none, hy, nbh);
    } else {
      hipLaunchKernelGGL(gsc_prepare_kernel<false>, dim3((unsigned)(nbc + nbd)), dim3(256), 0, st, a, seeds3, dx3, none, hy, nbc);
    }
    // thousands of seeds, narrow features, short rows: the dense product on the matrix cores; otherwise the per-seed gather
    // (measured at 8192 seeds x 2000 genes x 50, round 6: gather 0.10 ms + counts 0.07; the dense product 0.17 + 0.07 — its tiles wait
    // for their L2 round trip with one workgroup per CU (116 KB of LDS) — so the gather stays the default; DANCE_AMD_GRAPHSC_AGG=mfma)
    static const bool agg_mfma = getenv("DANCE_AMD_GRAPHSC_AGG") && !strcmp(getenv("DANCE_AMD_GRAPHSC_AGG"), "mfma");
    if (agg_mfma && B >= 1024 && F <= 64 && c->max_row_entries > 0 && c->max_row_entries <= AG_MAXE + 1) {
      const size_t lds = (size_t)(AG_KC * AG_LDA + AG_KC * AG_LDB + 2 * AG_R * AG_MAXE + 2 * AG_R) * 4;
      const int64_t ldg3 = drop3 ? F : c->ld_features;
      const bool v2 = F % 2 == 0 && ldg3 % 2 == 0 && (reinterpret_cast<uintptr_t>(c->features) & 7u) == 0;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gsc_aggregate_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gsc_aggregate_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
      }
      if (v2) hipLaunchKernelGGL(gsc_aggregate_mfma_kernel<true>, dim3((unsigned)dh::ceil_div(B, AG_R)), dim3(256), lds, st, a, seeds3, dx3);
      else hipLaunchKernelGGL(gsc_aggregate_mfma_kernel<false>, dim3((unsigned)dh::ceil_div(B, AG_R)), dim3(256), lds, st, a, seeds3, dx3);
    } else {
      hipLaunchKernelGGL(gsc_forward_kernel<true>, dim3((unsigned)B), dim3(256), 0, st, a, seeds3, dx3, dx3, (float*)nullptr);
    }
    return dh::check_launch(me);
  }
  // the mirror of W2 and the gene counters may be stale (an eager step in between, a first call): one launch each per call
  hipLaunchKernelGGL(ms_transpose_kernel, dim3((unsigned)dh::ceil_div((int64_t)E * H, 256)), dim3(256), 0, st, E, H, c->w2.param, w2t);
  if (dh::zero_async(a.count, (size_t)G * 4, st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: clearing the counters failed", me);
  const bool drop = c->dropout > 0.f;
  const int nb_count = (B + 3) / 4;
  const int nb_drop = drop ? (int)dh::ceil_div(2 * (((int64_t)G * F + 3) / 4), 256) : 0;
  const size_t dec_lds = (size_t)(2 * E + B) * sizeof(float);
  for (int64_t s = first_step; s < first_step + n_steps; ++s) {
    const uint64_t gstep = c->step0 + (uint64_t)s;
    const Drop dx = make_drop(c->dropout, c->seed, gstep), dd = make_drop(c->decoder_dropout, c->seed, gstep);
    const int64_t* seeds = c->seeds + s * B;
    hipLaunchKernelGGL(gsc_prepare_kernel<false>, dim3((unsigned)(nb_count + nb_drop)), dim3(256), 0, st, a, seeds, dx, sc, hy, nb_count);
    hipLaunchKernelGGL(gsc_forward_kernel<false>, dim3((unsigned)B), dim3(256), 0, st, a, seeds, dx, dd, c->z_out + s * (int64_t)B * E);
    hipLaunchKernelGGL(gsc_decoder_kernel, dim3((unsigned)B), dim3(256), dec_lds, st, a, dd, (float)pos_weight, (float)(norm / (b * b)));
    ga.loss_out = c->loss_out + s;
    launch_grad(ga, B, st);
  }
  return dh::check_launch(me);
}

// ---- scDeepSort ---------------------------------------------------------------------------------------------------------------------
namespace {
struct SdsLayout {
  size_t coef, h1, dh1, dlog, rowloss, total;
};
SdsLayout sds_layout(int64_t B, int64_t D, int64_t H, int64_t C) {
  SdsLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += a256(bytes);
    return at;
  };
  l.coef = take(16);
  l.h1 = take((size_t)B * H * 4);
  l.dh1 = take((size_t)B * H * 4);
  l.dlog = take((size_t)B * C * 4);
  l.rowloss = take((size_t)B * 4);
  l.total = o;
  return l;
}
}  // namespace

extern "C" int dh_scdeepsort_step_supported(int64_t batch, int64_t dim_in, int64_t hidden, int64_t n_classes) {
  // sds_loss keeps W2 [C, H + 1] and four rows in LDS; sds_neigh four features per thread
  return (batch >= 1 && batch <= 65536 && dim_in >= 1 && dim_in <= 1024 && hidden >= 1 && hidden <= 1024 && n_classes >= 1 && n_classes <= 64 &&
          (n_classes * (hidden + 1) + 4 * (hidden + 64)) * 4 <= 60 * 1024) ? 1 : 0;
}

extern "C" size_t dh_scdeepsort_step_workspace_bytes(int64_t batch, int64_t dim_in, int64_t hidden, int64_t n_classes) {
  if (!dh_scdeepsort_step_supported(batch, dim_in, hidden, n_classes)) return 0;
  return sds_layout(batch, dim_in, hidden, n_classes).total;
}

extern "C" int dh_scdeepsort_steps(const dh_scdeepsort_step_t* c, int64_t first_step, int64_t n_steps, dh_stream_t stream) {
  const char* me = "dh_scdeepsort_steps";
  if (!c) return dh::fail(DH_ERR_INVALID, "%s: null configuration", me);
  if (first_step < 0 || n_steps < 0) return dh::fail(DH_ERR_INVALID, "%s: negative step range", me);
  if (n_steps == 0) return DH_OK;
  if (!dh_scdeepsort_step_supported(c->batch, c->dim_in, c->hidden, c->n_classes))
    return dh::fail(DH_ERR_INVALID, "%s: batch %lld, %lld -> %lld -> %lld outside dh_scdeepsort_step_supported", me, (long long)c->batch, (long long)c->dim_in,
                    (long long)c->hidden, (long long)c->n_classes);
  if (c->n_genes < 0 || c->n_nodes <= 0 || c->n_nodes >= ((int64_t)1 << 31)) return dh::fail(DH_ERR_INVALID, "%s: bad node counts", me);
  if (!c->features || !c->labels || !c->seeds || !c->bad || !c->workspace) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (c->neigh_out && (!c->rowptr || !c->col || !c->val || !c->cell_id || !c->alpha)) return dh::fail(DH_ERR_INVALID, "%s: the aggregation needs the graph, cell_id and alpha", me);
  if (c->ld_features < c->dim_in) return dh::fail(DH_ERR_INVALID, "%s: ld_features < dim_in", me);
  if (!adam_ok(c->w1) || !adam_ok(c->b1) || !adam_ok(c->w2) || !adam_ok(c->b2)) return dh::fail(DH_ERR_INVALID, "%s: incomplete Adam state", me);
  if (c->dropout < 0.f || c->dropout >= 1.f) return dh::fail(DH_ERR_INVALID, "%s: dropout outside [0, 1)", me);
  if (c->phase < 0 || c->phase > 2 || (c->phase != 0 && (n_steps != 1 || !c->grads))) return dh::fail(DH_ERR_INVALID, "%s: phases 1 / 2 take one step and a gradient buffer", me);
  if (c->phase != 2 && !c->loss_out) return dh::fail(DH_ERR_INVALID, "%s: null output", me);
  const int B = (int)c->batch, D = (int)c->dim_in, H = (int)c->hidden, C = (int)c->n_classes;
  const SdsLayout l = sds_layout(B, D, H, C);
  if (c->workspace_bytes < l.total) return dh::fail(DH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", me, c->workspace_bytes, l.total);
  hipStream_t st = dh::as_stream(stream);
  char* ws = static_cast<char*>(c->workspace);
  SdsArgs a{};
  a.rowptr = c->rowptr; a.col = c->col; a.val = c->val; a.X = c->features; a.ldx = c->ld_features; a.n_nodes = c->n_nodes;
  a.x_bf16 = c->features_bf16 ? 1 : 0; a.G = (int)c->n_genes; a.B = B; a.D = D; a.H = H; a.C = C;
  a.cell_id = c->cell_id; a.labels = c->labels; a.alpha = c->alpha;
  a.W1 = c->w1.param; a.b1 = c->b1.param; a.W2 = c->w2.param; a.b2 = c->b2.param;
  a.coef = reinterpret_cast<float*>(ws + l.coef);
  a.h1 = reinterpret_cast<float*>(ws + l.h1);
  a.dh1 = reinterpret_cast<float*>(ws + l.dh1);
  a.dlog = reinterpret_cast<float*>(ws + l.dlog);
  a.rowloss = reinterpret_cast<float*>(ws + l.rowloss);
  a.neigh = c->neigh_out;
  a.bad = c->bad;
  const AdamHyper hy{c->lr, c->beta1, c->beta2, c->eps, c->weight_decay};
  StepCounters sc{};
  sc.step[0] = c->w1.step; sc.step[1] = c->b1.step; sc.step[2] = c->w2.step; sc.step[3] = c->b2.step;
  sc.n = 4;
  MsGradArgs ga{};
  float* g = c->phase ? c->grads : nullptr;
  const int64_t o_b1 = (int64_t)H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + (int64_t)C * H;
  set_job(ga.job[0], a.dh1, H, c->features, c->ld_features, H, D, B, c->w1, nullptr, g);      // dW1 = dH1^T dropout(X[seeds])
  ga.job[0].b_bf16 = a.x_bf16;
  ga.job[0].b_drop = 1;
  set_job(ga.job[1], nullptr, 0, a.dh1, H, 1, H, B, c->b1, nullptr, g ? g + o_b1 : nullptr);   // db1
  set_job(ga.job[2], a.dlog, C, a.h1, H, C, H, B, c->w2, nullptr, g ? g + o_w2 : nullptr);     // dW2 = dLogits^T h1
  set_job(ga.job[3], nullptr, 0, a.dlog, C, 1, C, B, c->b2, nullptr, g ? g + o_b2 : nullptr);  // db2
  ga.n_jobs = 4;
  place_tiles(ga);
  ga.adam = c->phase == 0;
  ga.hy = hy;
  ga.coef = a.coef;
  ga.rowloss_f = a.rowloss;
  ga.n_rowloss = B;
  ga.loss_div = 1.0;
  ga.loss_mul = 1.f;
  if (c->phase == 2) {
    int64_t longest = 0;
    for (int j = 0; j < ga.n_jobs; ++j) longest = std::max<int64_t>(longest, (int64_t)ga.job[j].M * ga.job[j].N);
    hipLaunchKernelGGL(ms_adam_kernel, dim3((unsigned)dh::ceil_div(longest, 256), (unsigned)ga.n_jobs), dim3(256), 0, st, ga);
    return dh::check_launch(me);
  }
  const int DP = std::min(pow2_at_least(D), 256);
  const size_t neigh_lds = (size_t)(256 / DP) * D * sizeof(float);
  // rows that start on 16-byte (fp32) / 8-byte (bf16) boundaries take the vector kernel
  const bool neigh_vec = D % 4 == 0 && c->ld_features % 4 == 0 && (reinterpret_cast<uintptr_t>(c->features) & 15u) == 0;
  const size_t loss_lds = (size_t)(C * (H + 1) + 4 * (H + 64)) * sizeof(float);
  // The discarded aggregation depends on nothing the step computes and nothing depends on it (gnn.py:90-92): it runs on a SIDE stream next to
  // the step's own chain — forked from the caller's stream at the call's start, joined at its end, so for the caller the call is still one
  // in-order piece of work on `stream`.  20 of a step's 60 us at batch 500 (profiles/r06h); DANCE_AMD_MINISTEP_SIDE_STREAM=0 keeps one stream.
  static hipStream_t side = nullptr;
  static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  static const bool want_side = !(getenv("DANCE_AMD_MINISTEP_SIDE_STREAM") && !strcmp(getenv("DANCE_AMD_MINISTEP_SIDE_STREAM"), "0"));
  hipStream_t nst = st;
  if (c->neigh_out && want_side) {
    if (!side && (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess)) {
      side = nullptr;
      (void)hipGetLastError();
    }
    if (side && hipEventRecord(ev_fork, st) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess) nst = side;
  }
  for (int64_t s = first_step; s < first_step + n_steps; ++s) {
    const Drop dz = make_drop(c->dropout, c->seed, c->step0 + (uint64_t)s);
    const int64_t* seeds = c->seeds + s * B;
    if (c->neigh_out) {
      if (neigh_vec) {
        const size_t lds = (size_t)4 * D * sizeof(float);
        const int nq = (D / 4 + 63) / 64;
#define DH_NEIGH(NQ)                                                                                                          \
  do {                                                                                                                        \
    if (a.x_bf16) hipLaunchKernelGGL((sds_neigh_vec_kernel<NQ, true>), dim3((unsigned)B), dim3(256), lds, nst, a, seeds);   \
    else hipLaunchKernelGGL((sds_neigh_vec_kernel<NQ, false>), dim3((unsigned)B), dim3(256), lds, nst, a, seeds);          \
  } while (0)
        if (nq <= 1) DH_NEIGH(1);
        else if (nq <= 2) DH_NEIGH(2);
        else DH_NEIGH(4);
#undef DH_NEIGH
      } else {
        hipLaunchKernelGGL(sds_neigh_kernel, dim3((unsigned)B), dim3(256), neigh_lds, nst, a, seeds);
      }
    }
    {
      const dim3 hg((unsigned)dh::ceil_div(H, MS_T), (unsigned)dh::ceil_div(B, MS_T));
      if (D <= 128) hipLaunchKernelGGL(sds_hidden_kernel<128>, hg, dim3(256), 0, st, a, seeds, dz, sc, hy);
      else hipLaunchKernelGGL(sds_hidden_kernel<256>, hg, dim3(256), 0, st, a, seeds, dz, sc, hy);
    }
    hipLaunchKernelGGL(sds_loss_kernel, dim3((unsigned)dh::ceil_div(B, 4)), dim3(256), loss_lds, st, a, seeds);
    ga.job[0].b_rows = seeds;
    ga.bdrop = dz;
    ga.loss_out = c->loss_out + s;
    launch_grad(ga, B, st);
  }
  if (nst != st && (hipEventRecord(ev_join, nst) != hipSuccess || hipStreamWaitEvent(st, ev_join, 0) != hipSuccess))
    return dh::fail(DH_ERR_LAUNCH, "%s: joining the side stream failed", me);
  return dh::check_launch(me);
}
