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

// knn_filter.hip: matrix-core filter + exact re-rank behind dh_knn_bruteforce_f32
int64_t knn_filter_sample_size(int64_t n);
int knn_filter_cap(int64_t n, int k);
void knn_filter_geometry(int64_t n, int64_t d, int64_t nq, int k, int* n_seg, int* seg);
int knn_filter_padded_d(int64_t d);
int64_t knn_filter_k3(int64_t d);
void knn_filter_sample(int64_t n, int64_t d, const float* X, int64_t ldx, int rs, float* Xs, hipStream_t st);
int knn_filter_launch(int64_t n, int64_t d, const float* X, int64_t ldx, const float* Xr, int64_t ldr, int64_t dr,
                      int64_t q_begin, int64_t nq, int k, const float* sample_d2, float* mean_ws, uint16_t* A2, uint16_t* B2,
                      float* norms, float* Rq, float* Cn, int32_t* counts, int32_t* surv, int32_t* out_idx, float* out_dist,
                      hipStream_t st);
size_t knn_filter_mean_floats(int64_t d);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace dh
