// Library-level entry points: version, error text, device probe.
#include "common.h"

namespace dh {

char* error_buffer() {
  static thread_local char buf[512] = "ok";
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace dh

extern "C" {

int dh_version(void) { return 100; /* 0.1.0 */ }

const char* dh_last_error_string(void) { return dh::error_buffer(); }

int dh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"


// ---- dh::zero_async: see common.h ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned char* __restrict__ p, size_t head, size_t words16, size_t tail_begin, size_t bytes) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (size_t i = tid; i < head; i += stride) p[i] = 0;                     // up to the first 16-byte boundary
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (size_t i = tid; i < words16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = tail_begin + tid; i < bytes; i += stride) p[i] = 0;
}
}  // namespace

hipError_t dh::zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if (!p) return hipErrorInvalidValue;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  size_t head = (16 - (a & 15u)) & 15u;
  if (head > bytes) head = bytes;
  const size_t words16 = (bytes - head) / 16, tail_begin = head + words16 * 16;
  const size_t work = words16 > 0 ? words16 : bytes;
  const unsigned grid = (unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, static_cast<unsigned char*>(p), head, words16, tail_begin, bytes);
  return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void zero2d_kernel(unsigned char* __restrict__ p, size_t pitch, size_t width, size_t rows) {
  const size_t total = width * rows;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) p[(i / width) * pitch + i % width] = 0;
}
}  // namespace

hipError_t dh::zero2d_async(void* p, size_t pitch, size_t width_bytes, size_t rows, hipStream_t st) {
  if (width_bytes == 0 || rows == 0) return hipSuccess;
  if (!p) return hipErrorInvalidValue;
  if (pitch == width_bytes) return dh::zero_async(p, width_bytes * rows, st);
  const size_t total = width_bytes * rows;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(zero2d_kernel, dim3(grid), dim3(256), 0, st, static_cast<unsigned char*>(p), pitch, width_bytes, rows);
  return hipGetLastError();
}
