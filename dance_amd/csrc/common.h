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

// knn_grid.hip: exact kNN of points in <= 3 dimensions by a uniform cell grid (DH_KNN_GRID)
bool knn_grid_supported(int64_t d, int k);
bool knn_grid_applies(int64_t n, int64_t d, int k);
size_t knn_grid_workspace_bytes(int64_t n);
int knn_grid_launch(int64_t n, int d, const float* X, int64_t ldx, int64_t q_begin, int64_t nq, int k, int32_t* out_idx, float* out_dist, void* workspace,
                    hipStream_t st);

// sage_bcm.hip: the two-waves-per-SIMD kernel pair (block-chunk-major plan of the graph + MFMA) behind the unsplit path of dh_sage_window_mfma
bool sage_bcm_fits(int64_t n_dst, int64_t n_cols, int64_t width, bool hbf16, const void* H, int64_t ldh, int64_t nnz);
size_t sage_bcm_plan_bytes(int64_t n_dst, int64_t n_cols, int64_t nnz);
size_t sage_bcm_prep_bytes(int64_t n_cols, int64_t width, bool hbf16);
size_t sage_bcm_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, bool hbf16, int64_t nnz);
int sage_bcm_plan(int64_t n_dst, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col, const float* w, void* plan,
                  hipStream_t st);
int sage_bcm_launch(int64_t n_dst, int64_t width, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                    const float* w, const float* colscale, const void* H, int64_t ldh, bool hb, void* neigh, int64_t ldn, bool ob,
                    int64_t nnz, const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha, int64_t n_genes,
                    const void* plan, void* workspace, hipStream_t st);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Zero `bytes` bytes at p with a KERNEL launch (lib.hip).  Never hipMemsetAsync in this library: inside a captured hipGraph a memset
// becomes a memset NODE, and on ROCm 7.2 / gfx950 such a node was replayed with another operation's fill pattern and extent after eager
// copies had run between two replays (round 5: a 4-byte "out[0] = 0" node filled a block's whole row-pointer array with 0x80 bytes and
// the next kernel stored 8 GB below its buffer — profiles/r05_replay_fault.md).  Kernel nodes carry their arguments with them.
hipError_t zero_async(void* p, size_t bytes, hipStream_t st);
// rows x width_bytes zeros at p with `pitch` bytes between row starts (hipMemset2DAsync's shape), again as a kernel
hipError_t zero2d_async(void* p, size_t pitch, size_t width_bytes, size_t rows, hipStream_t st);

}  // namespace dh
