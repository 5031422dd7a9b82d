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
    out_col[p] = (int32_t)lo;
    if (out_val) out_val[p] = val[e];
  }
}

// Small graphs — the message-flow block of a mini-batch (129 rows of ~200 entries, transposed once per training step inside a captured
// hipGraph, where the sort-based path is 12 dependent launches of ~5 us each) — in ONE workgroup, every phase parallel over the
// entries or the columns:
//   1. histogram of the columns in LDS, block-wide exclusive scan = out_rowptr;
//   2. every entry claims a slot of its column's segment through an LDS cursor — in arrival order, i.e. unordered;
//   3. one thread per column insertion-sorts its segment by source row (segments are a dozen entries; the stable sort's order is
//      "ascending source row", and entries of one source row never share a column except in a static block's padding row, whose
//      entries are interchangeable zeros).
// The first version placed the rows one after the other with a barrier per row: correct, but 129 barriers made it SLOWER than the 12
// launches it replaced (3.9 vs 3.3 s per 1M-cell graph-sc epoch).  Inconsistent input (a column >= n_cols, more entries than the
// caller's nnz) is skipped instead of written out of bounds; -DDH_TRANSPOSE_DEBUG reports it.
#ifndef DH_TRANSPOSE_SMALL_T
#define DH_TRANSPOSE_SMALL_T 1024
#endif
constexpr int SMALL_T = DH_TRANSPOSE_SMALL_T;
__global__ __launch_bounds__(SMALL_T) void csr_transpose_small_kernel(int n_rows, int n_cols, int nnz_cap, const int32_t* __restrict__ rowptr,
                                                                      const int32_t* __restrict__ col, const float* __restrict__ val,
                                                                      int32_t* __restrict__ out_rowptr, int32_t* __restrict__ out_col,
                                                                      float* __restrict__ out_val, int32_t* __restrict__ out_perm) {
  extern __shared__ int cursor[];  // [n_cols + 1] counts -> segment starts -> running cursors
  __shared__ int wave_tot[SMALL_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nnz = min(rowptr[n_rows], nnz_cap);  // never beyond the buffers the caller sized for nnz_cap entries
#ifdef DH_TRANSPOSE_DEBUG
  if (tid == 0 && rowptr[n_rows] != nnz_cap) printf("csr_transpose_small: rowptr[%d] = %d but the caller said nnz = %d\n", n_rows, rowptr[n_rows], nnz_cap);
#endif
  for (int i = tid; i <= n_cols; i += SMALL_T) cursor[i] = 0;
  __syncthreads();
  for (int e = tid; e < nnz; e += SMALL_T) {
    const int c = col[e];
    if ((unsigned)c < (unsigned)n_cols) atomicAdd(&cursor[c], 1);
#ifdef DH_TRANSPOSE_DEBUG
    else printf("csr_transpose_small: column %d at entry %d outside [0, %d) (nnz %d of cap %d, rows %d)\n", c, e, n_cols, nnz, nnz_cap, n_rows);
#endif
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
  int base = incl - sum;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  for (int i = b; i < e_; ++i) {
    const int c = cursor[i];
    cursor[i] = base;
    out_rowptr[i] = base;
    base += c;
  }
  __syncthreads();
  // unordered placement: a wave owns whole rows (row id = one binary search per entry avoided), lanes stride over the row's entries
  for (int r = wave; r < n_rows; r += SMALL_T / 64) {
    const int rs = rowptr[r], re = min(rowptr[r + 1], nnz);
    for (int e0 = rs; e0 < re; e0 += 64) {
      const int e = e0 + lane;
      const bool live = e < re;
      const int c = live ? col[e] : -1;
      if (!live || (unsigned)c >= (unsigned)n_cols) continue;
      // 64 entries of ONE column (a static block's padding row: thousands of zeros in column 0): one cursor bump for the chunk,
      // slots in entry order — the run arrives sorted and the insertion sort below passes over it in linear time
      const int c0 = __builtin_amdgcn_readfirstlane(c);
      const unsigned long long act = __ballot(true);
      int pos;
      if (__ballot(c == c0) == act) {
        const int lead = __ffsll((long long)act) - 1;
        int first = 0;
        if (lane == lead) first = atomicAdd(&cursor[c0], __popcll(act));
        first = __shfl(first, lead, 64);
        pos = first + __popcll(act & ((1ull << lane) - 1ull));
      } else {
        pos = atomicAdd(&cursor[c], 1);
      }
      if (pos >= nnz) {
#ifdef DH_TRANSPOSE_DEBUG
        printf("csr_transpose_small: cursor %d of column %d past nnz %d (row %d entry %d)\n", pos, c, nnz, r, e);
#endif
        continue;
      }
      out_col[pos] = r;
      out_perm[pos] = e;
      if (out_val) out_val[pos] = val[e];
    }
  }
  __syncthreads();  // (the block's own global writes are visible to the block after the barrier)
  // per-column insertion sort by (source row, entry position): the stable sort's order
  for (int c = tid; c < n_cols; c += SMALL_T) {
    const int s0 = out_rowptr[c], s1 = min(cursor[c], nnz);  // cursor[c] now = end of the column's segment
    for (int i = s0 + 1; i < s1; ++i) {
      const int r = out_col[i], e = out_perm[i];
      const float v = out_val ? out_val[i] : 0.f;
      int j = i - 1;
      while (j >= s0 && (out_col[j] > r || (out_col[j] == r && out_perm[j] > e))) {
        out_col[j + 1] = out_col[j];
        out_perm[j + 1] = out_perm[j];
        if (out_val) out_val[j + 1] = out_val[j];
        --j;
      }
      out_col[j + 1] = r;
      out_perm[j + 1] = e;
      if (out_val) out_val[j + 1] = v;
    }
  }
}
bool transpose_small_applies(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  static const bool on = getenv("DANCE_AMD_TRANSPOSE_SMALL") && getenv("DANCE_AMD_TRANSPOSE_SMALL")[0] == '1';  // round-5 switch, default off
  return on && n_rows <= 2048 && n_cols <= 12000 && nnz <= 262144;
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
#ifdef DH_TRANSPOSE_STATIC_LDS
    const size_t dyn_lds = 0;
#else
    const size_t dyn_lds = (size_t)(n_cols + 1) * sizeof(int);
#endif
    hipLaunchKernelGGL(csr_transpose_small_kernel, dim3(1), dim3(SMALL_T), dyn_lds, st, (int)n_rows, (int)n_cols, (int)nnz,
                       rowptr, col, val, out_rowptr, out_col, out_val, out_perm);
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
