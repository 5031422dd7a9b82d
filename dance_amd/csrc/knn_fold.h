// d <= 64 path of the kNN filter (knn_filter.hip): geometry shared with the workspace carve-up of knn.hip.
#pragma once
#include "common.h"

namespace dh {

struct KnnFoldGeom {
  int dp;              // features padded to a multiple of 8
  int K3;              // fp16 columns of an operand row: dp features + 6 threshold columns, whole 16-wide MFMA steps
  int G, H;            // rows grouped by r % G; the first H slots hold the classes that are multiples of G / H (knn_filter.hip)
  unsigned int qmagic; // ceil(2^16 / (G / H - 1))
  int n_pass;          // 1: everything; 2: every 16th row, the rest; 3: every 64th row, the other multiples of 8, the rest
  int64_t n1, n_pos;   // rows per residue class (ceil(n / G)); rows of B (G * n1 rounded up to whole LDS images; rows without a candidate are zero)
  int64_t S, stride0;  // the strided sample that gives the first thresholds: rows j * stride0, j < S (a subset of pass 1)
  int64_t tiles;       // 128-row candidate tiles of B
  int64_t t_begin[3], t_end[3], tps[3];  // tile range of each pass; tiles per candidate slice (grid.y)
  int n_seg[3], seg[3];                  // survivor list of a query: per pass n_seg segments (one per slice) of seg slots, passes in order
  int n_seg_total;
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
