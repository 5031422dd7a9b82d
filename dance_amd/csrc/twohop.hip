// Two-hop adjacency pattern for scHeteroNet's HetConv:
// dance/modules/single_modality/cell_type_annotation/scheteronet.py:507-539 (HeteroNet.init_adj) builds, next to the
// normalised one-hop adjacency, the pattern of  ((A A) - A) > 0  — reached by two edges and not "used up" by the direct edge —
// with torch_sparse's SpGEMM and scipy on the host.  Here it is a symbolic SpGEMM on the device (set-up, once per graph):
//
//   dh_csr_two_hop_count  : rowcnt[i] = sum_{j in N(i)} deg(j)                 (number of two-edge paths out of row i)
//   dh_csr_two_hop_expand : every path (i, j, c) becomes the 64-bit key (i << 32 | c); the keys are radix-sorted
//                           (rocPRIM) so equal (i, c) are adjacent; the first key of every run is flagged when
//                           #paths(i, c) - A[i, c] > 0, i.e. when (i, c) is not an edge of A or is reached by >= 2 paths
//                           (and, with drop_diag, c != i)
//   dh_csr_two_hop_compact: flagged keys -> CSR (row pointers by binary search over the sorted keys + the scanned flags)
//
// A is a 0/1 pattern in CSR with ascending, duplicate-free columns per row.  The host (kernels.csr_two_hop) sequences the
// three calls with dh_exclusive_scan_i32 in between (two size reads).
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void two_hop_count_kernel(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            int32_t* __restrict__ rowcnt, int32_t* __restrict__ overflow) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long c = 0;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e];
    c += rowptr[j + 1] - rowptr[j];
  }
  if (c > 0x7fffffffLL) {
    *overflow = 1;
    c = 0;
  }
  rowcnt[i] = (int32_t)c;
}

// one wavefront per row i: for every neighbour j (sequential), the lanes copy N(j)
__global__ __launch_bounds__(256) void two_hop_expand_kernel(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ offs, unsigned long long* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  int64_t o = offs[i];
  const unsigned long long hi = (unsigned long long)i << 32;
  for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int j = col[e];
    const int s = rowptr[j], t = rowptr[j + 1];
    for (int k = s + lane; k < t; k += 64) keys[o + (k - s)] = hi | (unsigned int)col[k];
    o += t - s;
  }
}

__device__ __forceinline__ bool in_row(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int i, int c) {
  int lo = rowptr[i], hi = rowptr[i + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < c) lo = mid + 1; else hi = mid;
  }
  return lo < rowptr[i + 1] && col[lo] == c;
}

__global__ __launch_bounds__(256) void two_hop_flag_kernel(int64_t total, const unsigned long long* __restrict__ keys,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int drop_diag,
                                                           int32_t* __restrict__ flags) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= total) return;
  const unsigned long long key = keys[k];
  int f = 0;
  if (k == 0 || keys[k - 1] != key) {  // first key of its run
    const int i = (int)(key >> 32), c = (int)(key & 0xffffffffu);
    const bool twice = (k + 1 < total) && keys[k + 1] == key;
    f = (twice || !in_row(rowptr, col, i, c)) && !(drop_diag && i == c);
  }
  flags[k] = f;
}

__global__ __launch_bounds__(256) void two_hop_compact_kernel(int64_t total, const unsigned long long* __restrict__ keys,
                                                              const int32_t* __restrict__ flags, const int32_t* __restrict__ pos,
                                                              int32_t* __restrict__ out_col) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= total || !flags[k]) return;
  out_col[pos[k]] = (int32_t)(keys[k] & 0xffffffffu);
}

// out_rowptr[i] = number of kept keys with row < i
__global__ __launch_bounds__(256) void two_hop_rowptr_kernel(int64_t n, int64_t total, const unsigned long long* __restrict__ keys,
                                                             const int32_t* __restrict__ pos, int32_t* __restrict__ out_rowptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  const unsigned long long bound = (unsigned long long)i << 32;
  int64_t lo = 0, hi = total;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < bound) lo = mid + 1; else hi = mid;
  }
  out_rowptr[i] = pos[lo];  // pos has total + 1 entries (exclusive scan)
}

int key_bits(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n && b < 31) ++b;
  return 32 + b;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t sort_temp(int64_t total, int64_t n) {
  size_t bytes = 0;
  unsigned long long* d = nullptr;
  if (rocprim::radix_sort_keys(nullptr, bytes, d, d, (size_t)total, 0, key_bits(n), (hipStream_t)0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return bytes;
}

}  // namespace

extern "C" int dh_csr_two_hop_count(int64_t n, const int32_t* rowptr, const int32_t* col, int32_t* rowcnt, int32_t* overflow,
                                    dh_stream_t stream) {
  if (n < 0) return dh::fail(DH_ERR_INVALID, "dh_csr_two_hop_count: negative size");
  if (n == 0) return DH_OK;
  if (!rowptr || !col || !rowcnt || !overflow) return dh::fail(DH_ERR_INVALID, "dh_csr_two_hop_count: null pointer");
  hipStream_t st = dh::as_stream(stream);
  if (dh::zero_async(overflow, sizeof(int32_t), st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "dh_csr_two_hop_count: memset failed");
  hipLaunchKernelGGL(two_hop_count_kernel, dim3((unsigned)dh::ceil_div(n, 256)), dim3(256), 0, st, n, rowptr, col, rowcnt, overflow);
  return dh::check_launch("dh_csr_two_hop_count");
}

extern "C" size_t dh_csr_two_hop_workspace_bytes(int64_t n, int64_t total) {
  if (n <= 0 || total <= 0) return 0;
  return 2 * align256((size_t)total * 8) + align256(sort_temp(total, n)) + 256;
}

// keys_sorted (uint64 [total], inside the workspace: returned pointer offset = 0) and flags [total]
extern "C" int dh_csr_two_hop_expand(int64_t n, int64_t total, const int32_t* rowptr, const int32_t* col, const int32_t* offs, int drop_diag,
                                     int32_t* flags, void* workspace, size_t workspace_bytes, dh_stream_t stream) {
  const char* me = "dh_csr_two_hop_expand";
  if (n < 0 || total < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (n == 0 || total == 0) return DH_OK;
  if (!rowptr || !col || !offs || !flags) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  if (total >= (int64_t)1 << 31) return dh::fail(DH_ERR_INVALID, "%s: more than 2^31 two-edge paths", me);
  if (!workspace || workspace_bytes < dh_csr_two_hop_workspace_bytes(n, total) || ((uintptr_t)workspace & 255))
    return dh::fail(DH_ERR_WORKSPACE, "%s: workspace too small or not 256-byte aligned", me);
  hipStream_t st = dh::as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  unsigned long long* sorted = reinterpret_cast<unsigned long long*>(ws);
  unsigned long long* raw = reinterpret_cast<unsigned long long*>(ws + align256((size_t)total * 8));
  void* temp = ws + 2 * align256((size_t)total * 8);
  size_t temp_bytes = sort_temp(total, n);
  hipLaunchKernelGGL(two_hop_expand_kernel, dim3((unsigned)dh::ceil_div(n, 4)), dim3(256), 0, st, n, rowptr, col, offs, raw);
  hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, raw, sorted, (size_t)total, 0, key_bits(n), st);
  if (e != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: radix sort: %s", me, hipGetErrorString(e));
  hipLaunchKernelGGL(two_hop_flag_kernel, dim3((unsigned)dh::ceil_div(total, 256)), dim3(256), 0, st, total, sorted, rowptr, col, drop_diag, flags);
  return dh::check_launch(me);
}

extern "C" int dh_csr_two_hop_compact(int64_t n, int64_t total, const int32_t* flags, const int32_t* pos, int32_t* out_rowptr,
                                      int32_t* out_col, const void* workspace, dh_stream_t stream) {
  const char* me = "dh_csr_two_hop_compact";
  if (n < 0 || total < 0) return dh::fail(DH_ERR_INVALID, "%s: negative size", me);
  if (!out_rowptr) return dh::fail(DH_ERR_INVALID, "%s: null out_rowptr", me);
  hipStream_t st = dh::as_stream(stream);
  if (total == 0 || n == 0) {
    if (dh::zero_async(out_rowptr, (size_t)(n + 1) * sizeof(int32_t), st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "%s: memset failed", me);
    return DH_OK;
  }
  if (!flags || !pos || !workspace) return dh::fail(DH_ERR_INVALID, "%s: null pointer", me);
  const unsigned long long* sorted = static_cast<const unsigned long long*>(workspace);
  if (out_col) hipLaunchKernelGGL(two_hop_compact_kernel, dim3((unsigned)dh::ceil_div(total, 256)), dim3(256), 0, st, total, sorted, flags, pos, out_col);
  hipLaunchKernelGGL(two_hop_rowptr_kernel, dim3((unsigned)dh::ceil_div(n + 1, 256)), dim3(256), 0, st, n, total, sorted, pos, out_rowptr);
  return dh::check_launch(me);
}
