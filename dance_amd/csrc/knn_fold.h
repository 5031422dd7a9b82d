// d <= 64 path of the kNN filter (knn_filter.hip): geometry shared with the workspace carve-up of knn.hip.
#pragma once
#include "common.h"

namespace dh {

struct KnnFoldGeom {
  int dp;              // features padded to a multiple of 8
  int K3;              // fp16 columns of an operand row: dp features + 6 threshold columns, whole 16-wide MFMA steps
  int G;               // pass 1 covers the rows r with r % G == 0 (G == 1: one pass over everything)
  int64_t n1, n_pos;   // rows per residue class (ceil(n / G)); operand rows of B (G * n1 rounded up to whole LDS images; rows without a candidate are zero)
  int64_t S, stride0;  // the strided sample that gives the first thresholds: rows j * stride0, j < S (a subset of pass 1)
  int64_t tiles, t1;   // 128-row candidate tiles of B: pass 1 = [0, t1), pass 2 = [t1, tiles)
  int64_t tps1, tps2;  // tiles per candidate slice (grid.y) of each pass
  int n_seg1, seg1;    // survivor list of a query: n_seg1 segments of seg1 slots (pass 1, one per slice) ...
  int n_seg2, seg2;    // ... followed by n_seg2 segments of seg2 slots (pass 2)
  int cap;             // slots per query
};

bool knn_fold_applies(int64_t d);
KnnFoldGeom knn_fold_geom(int64_t n, int64_t d, int64_t nq, int k);

// rows j * stride of X (j < S), zero-padded to `rs` columns
void knn_filter_sample_strided(int64_t S, int64_t stride, int64_t d, const float* X, int64_t ldx, int rs, float* Xs, hipStream_t st);

// Steps 2 and 3 for d <= 64; `tau` = out_dist holds the raw k-th sample distances in column k - 1 on entry.
// Xr: the zero-padded copy [n][ldr] the re-rank reads (dr columns, a multiple of 4).
int knn_fold_launch(const KnnFoldGeom& g, int64_t n, int64_t d, const float* X, int64_t ldx, const float* Xr, int64_t ldr, int64_t dr,
                    int64_t q_begin, int64_t nq, int k, float* mean_ws, unsigned int* maxabs, void* A2, void* B2, float* norms,
                    int32_t* counts, int32_t* surv, int32_t* out_idx, float* out_dist, hipStream_t st);

}  // namespace dh
