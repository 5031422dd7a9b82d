// Exact brute-force kNN (rows A11/A12/A13 of §8a).
//
// Parity contract ("bit-exact graph indices"): the squared distance of a pair is DEFINED as
//     d2 = (((0 + sq(x_0-y_0)) + sq(x_1-y_1)) + ...),  sq(u) = rn(u*u), every op a separate f32
// round-to-nearest operation in feature order — exactly what the numpy oracle evaluates — and the
// neighbours of a query are the k smallest (d2, index) pairs in lexicographic order (ties go to
// the lower index; the query itself is its own nearest neighbour with d2 = 0, as in sklearn and
// scanpy).  No |x|^2 - 2xy expansion: that is what makes the index lists reproducible bit for
// bit; the matrix cores are deliberately not used here.
//
// Mapping: one lane per query (256 queries per block).  Candidate rows are staged through LDS in
// tiles and broadcast to all lanes (same-address ds_read_b128, conflict-free); the query's own
// features sit in registers.  Each lane keeps its current k-th best (tau) in registers; a
// candidate that beats tau is inserted into the lane's sorted list (LDS for k <= 32, else the
// output arrays themselves).  For random data a query sees only ~k ln(N/k) insertions, so the
// kernel is bound by the 3 VALU ops per (pair, feature).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ void insert(const List& L, int k, int& cnt, float& tau_d, int& tau_i, float d2, int idx) {
  int pos = cnt < k ? cnt : k - 1;
  while (pos > 0) {
    const float pd = L.d[(pos - 1) * L.stride];
    const int pi = L.i[(pos - 1) * L.stride];
    if (!before(d2, idx, pd, pi)) break;
    L.d[pos * L.stride] = pd;
    L.i[pos * L.stride] = pi;
    --pos;
  }
  L.d[pos * L.stride] = d2;
  L.i[pos * L.stride] = idx;
  if (cnt < k) ++cnt;
  if (cnt == k) {
    tau_d = L.d[(k - 1) * L.stride];
    tau_i = L.i[(k - 1) * L.stride];
  }
}

__device__ __forceinline__ void finish(const List& L, bool lds_list, int k, int cnt, int64_t q_local, bool valid,
                                       int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  if (!valid) return;
  int32_t* oi = out_idx + q_local * k;
  float* od = out_dist + q_local * k;
  for (int s = 0; s < k; ++s) {
    if (s < cnt) {
      const float d2 = L.d[s * L.stride];
      const int idx = L.i[s * L.stride];
      oi[s] = idx;
      od[s] = (float)sqrt((double)d2);  // f64 sqrt then one rounding == correctly rounded f32 sqrt
    } else {  // fewer than k points exist
      oi[s] = -1;
      od[s] = __int_as_float(0x7f800000);
    }
  }
  (void)lds_list;
}

// d <= DCH: the whole query row lives in registers, candidates stream through LDS 64 at a time.
template <int DCH>
__global__ __launch_bounds__(QB) void knn_small_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                       int64_t q_begin, int64_t q_end, int k, bool lds_list,
                                                       int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  constexpr int CT = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cand = reinterpret_cast<float*>(smem);                    // [CT][DCH]
  float* ld = cand + CT * DCH;                                     // [k][QB] when lds_list
  int* li = reinterpret_cast<int*>(ld + (lds_list ? k * QB : 0));  // [k][QB]

  const int tid = threadIdx.x;
  const int64_t q_local = (int64_t)blockIdx.x * QB + tid;
  const int64_t q = q_begin + q_local;
  const bool valid = q < q_end;

  float x[DCH];
#pragma unroll
  for (int t = 0; t < DCH; ++t) x[t] = (valid && t < d) ? X[q * ldx + t] : 0.f;

  List L;
  if (lds_list) { L.d = ld + tid; L.i = li + tid; L.stride = QB; }
  else { L.d = out_dist + (valid ? q_local : 0) * k; L.i = out_idx + (valid ? q_local : 0) * k; L.stride = 1; }
  int cnt = 0, tau_i = 0x7fffffff;
  float tau_d = __int_as_float(0x7f800000);

  for (int64_t c0 = 0; c0 < n; c0 += CT) {
    __syncthreads();
    for (int idx = tid; idx < CT * DCH; idx += QB) {
      const int c = idx / DCH, t = idx % DCH;
      cand[idx] = (c0 + c < n && t < d) ? X[(c0 + c) * ldx + t] : 0.f;
    }
    __syncthreads();
    const int lim = (int)min((int64_t)CT, n - c0);
    for (int c = 0; c < lim; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < DCH; t += 4) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(cand + c * DCH + t);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float diff = __fsub_rn(x[t + u], y[u]);
          acc = __fadd_rn(acc, __fmul_rn(diff, diff));
        }
      }
      const int idx = (int)(c0 + c);
      if (valid && (cnt < k || before(acc, idx, tau_d, tau_i))) insert(L, k, cnt, tau_d, tau_i, acc, idx);
    }
  }
  finish(L, lds_list, k, cnt, q_local, valid, out_idx, out_dist);
}

// d > 64: features in chunks of 16; 16 candidates per tile with one register accumulator each.
__global__ __launch_bounds__(QB) void knn_big_kernel(int64_t n, int64_t d, const float* __restrict__ X, int64_t ldx,
                                                     int64_t q_begin, int64_t q_end, int k, bool lds_list,
                                                     int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  constexpr int CT = 16, DCH = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* cand = reinterpret_cast<float*>(smem);  // [CT][DCH]
  float* ld = cand + CT * DCH;
  int* li = reinterpret_cast<int*>(ld + (lds_list ? k * QB : 0));

  const int tid = threadIdx.x;
  const int64_t q_local = (int64_t)blockIdx.x * QB + tid;
  const int64_t q = q_begin + q_local;
  const bool valid = q < q_end;
  const float* xq = X + (valid ? q : 0) * ldx;

  List L;
  if (lds_list) { L.d = ld + tid; L.i = li + tid; L.stride = QB; }
  else { L.d = out_dist + (valid ? q_local : 0) * k; L.i = out_idx + (valid ? q_local : 0) * k; L.stride = 1; }
  int cnt = 0, tau_i = 0x7fffffff;
  float tau_d = __int_as_float(0x7f800000);

  for (int64_t c0 = 0; c0 < n; c0 += CT) {
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    for (int64_t t0 = 0; t0 < d; t0 += DCH) {
      __syncthreads();
      {
        const int c = tid / DCH, t = tid % DCH;  // 256 threads == CT * DCH
        cand[tid] = (c0 + c < n && t0 + t < d) ? X[(c0 + c) * ldx + t0 + t] : 0.f;
      }
      float x[DCH];
#pragma unroll
      for (int t = 0; t < DCH; ++t) x[t] = (t0 + t < d) ? xq[t0 + t] : 0.f;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int t = 0; t < DCH; t += 4) {
          const f32x4 y = *reinterpret_cast<const f32x4*>(cand + c * DCH + t);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float diff = __fsub_rn(x[t + u], y[u]);
            acc[c] = __fadd_rn(acc[c], __fmul_rn(diff, diff));
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int64_t ci = c0 + c;
      if (valid && ci < n && (cnt < k || before(acc[c], (int)ci, tau_d, tau_i)))
        insert(L, k, cnt, tau_d, tau_i, acc[c], (int)ci);
    }
  }
  finish(L, lds_list, k, cnt, q_local, valid, out_idx, out_dist);
}

}  // namespace

extern "C" int dh_knn_bruteforce_f32(int64_t n, int64_t d, const float* X, int64_t ldx, int64_t q_begin,
                                     int64_t q_end, int k, int32_t* out_idx, float* out_dist, dh_stream_t stream) {
  if (n < 0 || d < 0 || k < 0) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: negative size");
  if (q_begin < 0 || q_end > n || q_begin > q_end)
    return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: bad query range [%lld, %lld)", (long long)q_begin, (long long)q_end);
  if (q_end == q_begin || k == 0) return DH_OK;
  if (!X || !out_idx || !out_dist) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: null pointer");
  if (ldx < d) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: ldx < d");
  if (n >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "dh_knn_bruteforce_f32: n >= 2^31");
  hipStream_t st = dh::as_stream(stream);
  const bool lds_list = k <= KLDS;
  const size_t list_bytes = lds_list ? (size_t)k * QB * 8 : 0;
  dim3 grid((unsigned)dh::ceil_div(q_end - q_begin, QB)), block(QB);
#define DH_KNN_SMALL(DCH)                                                                            \
  hipLaunchKernelGGL(knn_small_kernel<DCH>, grid, block, 64 * DCH * sizeof(float) + list_bytes, st, n, d, X, ldx, \
                     q_begin, q_end, k, lds_list, out_idx, out_dist)
  if (d <= 4) DH_KNN_SMALL(4);
  else if (d <= 16) DH_KNN_SMALL(16);
  else if (d <= 32) DH_KNN_SMALL(32);
  else if (d <= 64) DH_KNN_SMALL(64);
  else
    hipLaunchKernelGGL(knn_big_kernel, grid, block, 16 * 16 * sizeof(float) + list_bytes, st, n, d, X, ldx, q_begin,
                       q_end, k, lds_list, out_idx, out_dist);
#undef DH_KNN_SMALL
  return dh::check_launch("dh_knn_bruteforce_f32");
}
