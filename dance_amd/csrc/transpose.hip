// Deterministic CSR transpose (CSR of A^T) for the backward SpMM.
//
// The backward pass of Y = A Z is dZ = A^T dY.  Instead of scattering with float atomics
// (non-deterministic summation order), the transposed CSR is built once per graph and the
// same gather SpMM kernel is reused.  Graph set-up, not the per-epoch hot loop:
//   1. stable LSD radix sort of (key = column, value = nnz position) — rocPRIM device sort,
//      the ROCm library primitive for exactly this; stable + input positions ascending makes
//      every output row ordered by source row, i.e. bit-identical to scipy's tocsc();
//   2. out_rowptr[j] = lower_bound(sorted columns, j)            (hand-written, one thread per j)
//   3. out_col[p] = row owning nnz position perm[p] (upper_bound on rowptr), out_val gathered.
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void iota_kernel(int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (int32_t)i;
}

// out_rowptr[j] = number of sorted keys < j, j in [0, n_cols]
__global__ __launch_bounds__(256) void rowptr_from_sorted_kernel(int64_t n_cols, int64_t nnz,
                                                                 const int32_t* __restrict__ keys,
                                                                 int32_t* __restrict__ out_rowptr) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > n_cols) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < (int32_t)j) lo = mid + 1; else hi = mid;
  }
  out_rowptr[j] = (int32_t)lo;
}

__global__ __launch_bounds__(256) void gather_transposed_kernel(int64_t n_rows, int64_t nnz,
                                                                const int32_t* __restrict__ rowptr,
                                                                const float* __restrict__ val,
                                                                const int32_t* __restrict__ perm,
                                                                int32_t* __restrict__ out_col,
                                                                float* __restrict__ out_val) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * 256) {
    const int32_t e = perm[p];
    // row i with rowptr[i] <= e < rowptr[i+1]  (skips empty rows)
    int64_t lo = 0, hi = n_rows;  // invariant: rowptr[lo] <= e, rowptr[hi] > e
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] <= e) lo = mid; else hi = mid;
    }
    // entries behind the last of the n_rows rows (a static block's padding tail when the caller transposes its real rows only:
    // CSRGraph.t_rows) carry no information: they get value 0 AND row n_rows — the block's padding row, whose operand row is zero —
    // instead of being clamped onto the last real row, where 0 * (a non-finite gradient row) would make a NaN
    const bool pad = e >= rowptr[n_rows];
    out_col[p] = pad ? (int32_t)n_rows : (int32_t)lo;
    if (out_val) out_val[p] = pad ? 0.f : val[e];
  }
}

// Small graphs — the message-flow block of a mini-batch (128 rows of ~200 entries, transposed once per training step inside a captured
// hipGraph, where the sort-based path is 12 dependent launches of ~5 us each) — in TWO launches:
//   csr_transpose_small_kernel (ONE workgroup of 1024): histogram of the columns in LDS, block-wide exclusive scan = out_rowptr, then
//       every entry claims a slot of its column's segment through an LDS cursor — in arrival order, i.e. unordered.  Entries are dealt
//       to the threads 16 at a time with all 16 loads issued before the first use: one workgroup hides memory latency by depth or not
//       at all (the versions that walked rows paid a round trip per row and chunk: ~100 us);
//   csr_segment_sort_kernel (one WAVEFRONT per column, the whole chip): the segment's keys (source row, entry position) go into LDS,
//       every lane ranks its entries by counting the smaller keys (a broadcast LDS read per comparison), and the segment is rewritten
//       in rank order = the stable sort's order.  Sorted segments are recognised in one pass and left alone.
// History (profiles/r05_replay_fault.md): v1 placed the rows one after the other with a barrier per row (3.9 s per 1M-cell graph-sc
// epoch against 3.3 s for the sort path), v2 sorted the segments inside the single workgroup by insertion (4.4 s), v3 split the sort off
// but still walked rows (3.9 s).  Inconsistent input (a column >= n_cols, more entries than the caller's nnz) is skipped instead of
// written out of bounds.
constexpr int SMALL_T = 1024, SMALL_U = 16, SMALL_MAX_ROWS = 2048;
__global__ __launch_bounds__(SMALL_T) void csr_transpose_small_kernel(int n_rows, int n_cols, int nnz_cap, const int32_t* __restrict__ rowptr,
                                                                      const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                      int32_t* __restrict__ out_rowptr, int32_t* __restrict__ out_col,
                                                                      float* __restrict__ out_val, int32_t* __restrict__ out_perm) {
  extern __shared__ int cursor[];  // [n_cols + 1] counts -> segment starts -> running cursors
  __shared__ int rp[SMALL_MAX_ROWS + 1];
  __shared__ int wave_tot[SMALL_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i <= n_rows; i += SMALL_T) rp[i] = rowptr[i];
  for (int i = tid; i <= n_cols; i += SMALL_T) cursor[i] = 0;
  __syncthreads();
  const int nnz = min(rp[n_rows], nnz_cap);  // never beyond the buffers the caller sized for nnz_cap entries
  for (int base = 0; base < nnz; base += SMALL_T * SMALL_U) {
    int c[SMALL_U];
#pragma unroll
    for (int u = 0; u < SMALL_U; ++u) {
      const int e = base + u * SMALL_T + tid;
      c[u] = e < nnz ? col[e] : -1;
    }
#pragma unroll
    for (int u = 0; u < SMALL_U; ++u)
      if ((unsigned)c[u] < (unsigned)n_cols) atomicAdd(&cursor[c[u]], 1);
  }
  __syncthreads();
  // exclusive scan of cursor[0 .. n_cols]: a contiguous chunk per thread, then the threads' sums across the block
  const int per = (n_cols + 1 + SMALL_T - 1) / SMALL_T;
  const int b = tid * per, e_ = min(n_cols + 1, b + per);
  int sum = 0;
  for (int i = b; i < e_; ++i) sum += cursor[i];
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int base0 = incl - sum;
  for (int w = 0; w < wave; ++w) base0 += wave_tot[w];
  for (int i = b; i < e_; ++i) {
    const int cnt = cursor[i];
    cursor[i] = base0;
    out_rowptr[i] = base0;
    base0 += cnt;
  }
  __syncthreads();
  for (int base = 0; base < nnz; base += SMALL_T * SMALL_U) {
    int c[SMALL_U];
    float v[SMALL_U];
#pragma unroll
    for (int u = 0; u < SMALL_U; ++u) {
      const int e = base + u * SMALL_T + tid;
      c[u] = e < nnz ? col[e] : -1;
      v[u] = (e < nnz && val) ? val[e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SMALL_U; ++u) {
      if ((unsigned)c[u] >= (unsigned)n_cols) continue;
      const int e = base + u * SMALL_T + tid;
      int lo = 0, hi = n_rows;  // the row r with rp[r] <= e < rp[r + 1] (skips empty rows)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rp[mid] <= e) lo = mid; else hi = mid;
      }
      const int pos = atomicAdd(&cursor[c[u]], 1);
      if (pos >= nnz) continue;
      out_col[pos] = lo;
      out_perm[pos] = e;
      if (out_val) out_val[pos] = v[u];
    }
  }
}

// One wavefront per column: rank sort of the segment by (source row, entry position).  Segments of up to SEG_LDS entries are
// ranked and rewritten through LDS; longer ones (a hub column of a general graph, never a mini-batch block of <= 2048 rows) by a
// serial insertion — correct, slow.
constexpr int SEG_LDS = 2048;
__global__ __launch_bounds__(64) void csr_segment_sort_kernel(int n_cols, const int32_t* __restrict__ rowptr_t, int32_t* __restrict__ out_col,
                                                              float* __restrict__ out_val, int32_t* __restrict__ out_perm) {
  __shared__ unsigned long long keys[SEG_LDS], sorted[SEG_LDS];
  __shared__ float sval[SEG_LDS];
  const int lane = threadIdx.x;
  const int c = blockIdx.x;
  if (c >= n_cols) return;
  const int s0 = rowptr_t[c], s1 = rowptr_t[c + 1], len = s1 - s0;
  if (len <= 1) return;
  auto key_at = [&](int i) -> unsigned long long { return ((unsigned long long)(unsigned)out_col[s0 + i] << 32) | (unsigned)out_perm[s0 + i]; };
  if (len > SEG_LDS) {
    if (lane == 0) {
      for (int i = 1; i < len; ++i) {
        const int r = out_col[s0 + i], e = out_perm[s0 + i];
        const float vv = out_val ? out_val[s0 + i] : 0.f;
        int j = i - 1;
        while (j >= 0 && (out_col[s0 + j] > r || (out_col[s0 + j] == r && out_perm[s0 + j] > e))) {
          out_col[s0 + j + 1] = out_col[s0 + j];
          out_perm[s0 + j + 1] = out_perm[s0 + j];
          if (out_val) out_val[s0 + j + 1] = out_val[s0 + j];
          --j;
        }
        out_col[s0 + j + 1] = r;
        out_perm[s0 + j + 1] = e;
        if (out_val) out_val[s0 + j + 1] = vv;
      }
    }
    return;
  }
  for (int i = lane; i < len; i += 64) keys[i] = key_at(i);
  __syncthreads();
  bool ok = true;
  for (int i = lane; i + 1 < len; i += 64) ok = ok && keys[i] <= keys[i + 1];
  if (__ballot(!ok) == 0ull) return;   // sorted already
  for (int i = lane; i < len; i += 64) {
    const unsigned long long k = keys[i];
    int rk = 0;
    for (int j = 0; j < len; ++j) rk += keys[j] < k ? 1 : 0;   // keys are distinct (distinct entry positions)
    sorted[rk] = k;
    sval[rk] = out_val ? out_val[s0 + i] : 0.f;
  }
  __syncthreads();
  for (int i = lane; i < len; i += 64) {
    const unsigned long long k = sorted[i];
    out_col[s0 + i] = (int)(k >> 32);
    out_perm[s0 + i] = (int)(unsigned)k;
    if (out_val) out_val[s0 + i] = sval[i];
  }
}
bool transpose_small_applies(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  static const bool on = getenv("DANCE_AMD_TRANSPOSE_SMALL") && getenv("DANCE_AMD_TRANSPOSE_SMALL")[0] == '1';  // round-5 switch, default off
  return on && n_rows <= SMALL_MAX_ROWS && n_cols <= 12000 && nnz <= 262144;
}

int end_bit_for(int64_t n_cols) {
  int b = 1;
  while (((int64_t)1 << b) < n_cols && b < 31) ++b;
  return b;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp_bytes(int64_t nnz, int64_t n_cols) {
  size_t bytes = 0;
  int32_t* dummy = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, dummy, dummy, dummy, dummy, (size_t)nnz, 0,
                                           end_bit_for(n_cols), (hipStream_t)0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return bytes;
}

}  // namespace

extern "C" size_t dh_csr_transpose_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  (void)n_rows;
  if (nnz <= 0) return 0;
  return 2 * align256((size_t)nnz * sizeof(int32_t)) + align256(sort_temp_bytes(nnz, n_cols)) + 256;
}

extern "C" int dh_csr_transpose(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                                const int32_t* col, const float* val, int32_t* out_rowptr,
                                int32_t* out_col, float* out_val, int32_t* out_perm, void* workspace,
                                size_t workspace_bytes, dh_stream_t stream) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0) return dh::fail(DH_ERR_INVALID, "dh_csr_transpose: negative size");
  if (!out_rowptr) return dh::fail(DH_ERR_INVALID, "dh_csr_transpose: null out_rowptr");
  if ((val == nullptr) != (out_val == nullptr))
    return dh::fail(DH_ERR_INVALID, "dh_csr_transpose: val and out_val must both be given or both NULL");
  hipStream_t st = dh::as_stream(stream);
  if (nnz == 0) {
    if (dh::zero_async(out_rowptr, (size_t)(n_cols + 1) * sizeof(int32_t), st) != hipSuccess)
      return dh::fail(DH_ERR_LAUNCH, "dh_csr_transpose: memset failed");
    return DH_OK;
  }
  if (!rowptr || !col || !out_col || !out_perm)
    return dh::fail(DH_ERR_INVALID, "dh_csr_transpose: null pointer");
  if (transpose_small_applies(n_rows, n_cols, nnz)) {
    hipLaunchKernelGGL(csr_transpose_small_kernel, dim3(1), dim3(SMALL_T), (size_t)(n_cols + 1) * sizeof(int), st, (int)n_rows, (int)n_cols, (int)nnz,
                       rowptr, col, val, out_rowptr, out_col, out_val, out_perm);
    hipLaunchKernelGGL(csr_segment_sort_kernel, dim3((unsigned)n_cols), dim3(64), 0, st, (int)n_cols, out_rowptr, out_col, out_val, out_perm);
    return dh::check_launch("dh_csr_transpose");
  }
  const size_t need = dh_csr_transpose_workspace_bytes(n_rows, n_cols, nnz);
  if (!workspace || workspace_bytes < need)
    return dh::fail(DH_ERR_WORKSPACE, "dh_csr_transpose: workspace %zu < %zu bytes", workspace_bytes, need);

  char* ws = static_cast<char*>(workspace);
  ws = reinterpret_cast<char*>(align256(reinterpret_cast<size_t>(ws)));
  int32_t* keys_sorted = reinterpret_cast<int32_t*>(ws);
  ws += align256((size_t)nnz * sizeof(int32_t));
  int32_t* positions = reinterpret_cast<int32_t*>(ws);
  ws += align256((size_t)nnz * sizeof(int32_t));
  size_t temp_bytes = sort_temp_bytes(nnz, n_cols);

  const unsigned g = (unsigned)(dh::ceil_div(nnz, 256) < 8192 ? dh::ceil_div(nnz, 256) : 8192);
  hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(256), 0, st, nnz, positions);
  hipError_t e = rocprim::radix_sort_pairs(ws, temp_bytes, col, keys_sorted, positions, out_perm, (size_t)nnz, 0,
                                           end_bit_for(n_cols), st);
  if (e != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "dh_csr_transpose: radix sort: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3((unsigned)dh::ceil_div(n_cols + 1, 256)), dim3(256), 0, st,
                     n_cols, nnz, keys_sorted, out_rowptr);
  hipLaunchKernelGGL(gather_transposed_kernel, dim3(g), dim3(256), 0, st, n_rows, nnz, rowptr, val, out_perm,
                     out_col, out_val);
  return dh::check_launch("dh_csr_transpose");
}
