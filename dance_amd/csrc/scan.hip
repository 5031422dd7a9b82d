// Exclusive prefix sum of per-row counts -> CSR row pointers (graph set-up plumbing; rocPRIM device scan).
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {
size_t scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int32_t* dummy = nullptr;
  if (rocprim::inclusive_scan(nullptr, bytes, dummy, dummy, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return bytes;
}
}  // namespace

extern "C" size_t dh_exclusive_scan_i32_workspace_bytes(int64_t n) { return n > 0 ? scan_temp_bytes(n) + 256 : 0; }

// out has n + 1 entries: out[0] = 0, out[i + 1] = in[0] + ... + in[i]
extern "C" int dh_exclusive_scan_i32(int64_t n, const int32_t* in, int32_t* out, void* workspace,
                                     size_t workspace_bytes, dh_stream_t stream) {
  if (n < 0) return dh::fail(DH_ERR_INVALID, "dh_exclusive_scan_i32: negative size");
  if (!out) return dh::fail(DH_ERR_INVALID, "dh_exclusive_scan_i32: null out");
  hipStream_t st = dh::as_stream(stream);
  // out[0] = 0 by a KERNEL, not hipMemsetAsync: inside a captured hipGraph the 4-byte memset NODE was replayed (ROCm 7.2, gfx950) as a
  // fill of the whole row-pointer array with the byte 0x80 on some replays — StaticCellBlock's row pointers then read 0x80808080,
  // negative offsets slipped past the "fits in e_max" test of the fill kernel, and its stores faulted 8 GB below the buffer
  // (round 5's hunt: scripts/rebuild_graph_check.py, profiles/r05_replay_fault.md).  A kernel node has no such failure mode.
  if (dh::zero_async(out, sizeof(int32_t), st) != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "dh_exclusive_scan_i32: zeroing out[0] failed");
  if (n == 0) return DH_OK;
  if (!in) return dh::fail(DH_ERR_INVALID, "dh_exclusive_scan_i32: null in");
  size_t temp = scan_temp_bytes(n);
  if (!workspace || workspace_bytes < temp) return dh::fail(DH_ERR_WORKSPACE, "dh_exclusive_scan_i32: workspace %zu < %zu", workspace_bytes, temp);
  hipError_t e = rocprim::inclusive_scan(workspace, temp, in, out + 1, (size_t)n, rocprim::plus<int32_t>(), st);
  if (e != hipSuccess) return dh::fail(DH_ERR_LAUNCH, "dh_exclusive_scan_i32: %s", hipGetErrorString(e));
  return DH_OK;
}
