// Shared host-side helpers for libdancehip.so (gfx950 only; no dual CUDA/HIP paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dance_hip.h"

namespace dh {

constexpr int kWave = 64;  // CDNA4 wavefront

// thread-local error text behind dh_last_error_string()
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DH_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DH_OK;
}

inline hipStream_t as_stream(dh_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// gemm_skinny.hip: narrow-layer (K, N <= 64, many rows) GEMM used by dh_gemm_f32
bool skinny_applies(int64_t M, int64_t N, int64_t K, int trans_a);
int skinny_launch(int64_t M, int64_t N, int64_t K, int trans_b, const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, int accumulate, hipStream_t st);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace dh
