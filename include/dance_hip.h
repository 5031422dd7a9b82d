/*
 * dance_hip.h — C ABI of libdancehip.so, the MI355X (gfx950) kernels behind DANCE's GNN
 * message-passing hot path.
 *
 * The reference (OmicsML/dance) has no FFI of its own: on this path its Python layers drop
 * straight into third-party native kernels (torch sparse / DGL / numba / scanpy / sklearn).
 * Each entry point below names the reference call site whose native work it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes binding a DANCE
 * maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host; the library never owns
 *    caller memory and allocates nothing persistent (workspaces are passed in, their sizes
 *    come from the *_workspace_bytes queries);
 *  - dense matrices are row-major with an explicit leading dimension in ELEMENTS;
 *  - CSR graphs use int32 row pointers and int32 column indices (nnz < 2^31), rows = the
 *    DESTINATION nodes of message passing, columns = the SOURCE nodes;
 *  - every launcher takes the hipStream_t to enqueue on (as void*), is asynchronous and
 *    returns 0 on success or a negative dh_status; dh_last_error_string() describes the last
 *    failure on the calling thread.  Nothing throws across the ABI.
 */
#ifndef DANCE_HIP_H
#define DANCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dh_stream_t; /* hipStream_t */

#if defined(DH_BUILDING)
#define DH_API __attribute__((visibility("default")))
#else
#define DH_API
#endif

enum dh_status {
  DH_OK = 0,
  DH_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, unsupported combo) */
  DH_ERR_LAUNCH = -2,    /* hipGetLastError() after a launch */
  DH_ERR_WORKSPACE = -3, /* workspace too small */
  DH_ERR_NO_DEVICE = -4, /* no gfx950 device visible */
  DH_ERR_COMM = -5       /* RCCL missing or a collective failed (dh_comm_*) */
};

enum dh_act { DH_ACT_NONE = 0, DH_ACT_RELU = 1 };
enum dh_reduce { DH_REDUCE_SUM = 0, DH_REDUCE_MEAN = 1 };
enum dh_knn_algo { DH_KNN_AUTO = 0, DH_KNN_SCAN = 1, DH_KNN_FILTER = 2, DH_KNN_GRID = 3 };
enum dh_dtype { DH_DTYPE_F32 = 0, DH_DTYPE_BF16 = 1 };
enum dh_metric { DH_METRIC_EUCLIDEAN = 0, DH_METRIC_PEARSON = 1, DH_METRIC_SPEARMAN = 2 };

/* ---- library ---------------------------------------------------------------------------- */
DH_API int dh_version(void);                     /* 10000*major + 100*minor + patch                 */
DH_API const char* dh_last_error_string(void);   /* thread-local, never NULL                        */
DH_API int dh_device_count(void);                /* number of visible HIP devices (0 on a CPU box)  */

/* ---- K1/K2/K5: CSR SpMM with fused epilogue ------------------------------------------------
 * Y[i,:] = act( rowscale[i] * reduce_{e in row i} ( val[e] * colscale[col[e]] * Z[col[e],:] ) + bias )
 * Replaces torch.spmm(adj, support) at dance/modules/single_modality/clustering/scdsc.py:498 and
 * dance/modules/spatial/spatial_domain/spagcn.py:359 (+bias :360-361, relu scdsc.py:499-500),
 * its autograd transpose product (run on the transposed CSR), and DGL's
 * update_all(u_mul_e, sum|mean) with the GraphConv degree norms at
 * dance/modules/single_modality/clustering/graphsc.py:444-449,462-476.
 * val / rowscale / colscale / bias may be NULL (= 1, 1, 1, 0).  reduce=MEAN divides by the row's
 * edge count (0 for an empty row, DGL fn.mean semantics).                                      */
DH_API int dh_spmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t width,
                    const int32_t* rowptr, const int32_t* col, const float* val,
                    const float* rowscale, const float* colscale,
                    const float* Z, int64_t ldz, float* Y, int64_t ldy,
                    const float* bias, int act, int reduce, dh_stream_t stream);

/* SpMM with the layer's ReLU fused on both sides, so autograd's ReluBackward (G = dY * [Y > 0],
 * scdsc.py:499-500 + :286-288) never touches HBM:
 *   forward : as dh_spmm_csr_f32(act) and, if out_mask != NULL, records the sign mask of Y
 *             (dh_relu_mask_bytes(n_rows, width) bytes: per row a little-endian bitmap, bit c % 32 of 32-bit word c / 32 = [Y(row, c) > 0]);
 *   backward: called on the CSR of A^T with Z = dY and in_mask = the recorded mask: gathered rows are masked
 *             on the fly, i.e. it returns A^T (dY * [Y > 0]).
 * Requires width % 128 == 0 and 16-byte aligned rows (dh_relu_mask_bytes returns 0 otherwise; callers then use
 * dh_relu_backward_f32 + dh_spmm_csr_f32).  in_mask is indexed by the gathered column id; rows of the operand that carry
 * no mask (halo rows received already masked) get all-ones words.
 * The _rows variants compute only the n_list rows named in row_ids (NULL: rows [0, n_list)); rowptr, rowscale, Y and
 * out_mask are indexed by the row id itself.  The sharded layer uses them to run the rows that need no remote operand
 * while the halo exchange is in flight.  dh_gather_rows_f32 packs out[i,:] = X[idx[i],:] (optionally times the ReLU sign
 * mask of X's rows) into a contiguous send buffer.                                                   */
DH_API size_t dh_relu_mask_bytes(int64_t n_rows, int64_t width);
DH_API int dh_spmm_csr_relu_f32(int64_t n_rows, int64_t n_cols, int64_t width,
                         const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* Z, int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                         void* out_mask, const void* in_mask, dh_stream_t stream);
DH_API int dh_spmm_csr_rows_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width,
                         const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* rowscale, const float* colscale,
                         const float* Z, int64_t ldz, float* Y, int64_t ldy,
                         const float* bias, int act, int reduce, dh_stream_t stream);
DH_API int dh_spmm_csr_relu_rows_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width,
                         const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* Z, int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                         void* out_mask, const void* in_mask, dh_stream_t stream);
/* Column slices [slice_begin, slice_end) (units of 128 columns) of dh_spmm_csr_relu_rows_f32 over a width-wide layer; Z, Y,
 * bias and both masks are those of the WHOLE layer.  The pipelined layer (dance_amd/autograd.py) aggregates slice c on one
 * stream while the GEMM producing slice c + 1 (torch.mm, scdsc.py:497) runs on another.                              */
DH_API int dh_spmm_csr_relu_slices_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width,
                         int64_t slice_begin, int64_t slice_end,
                         const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* Z, int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                         void* out_mask, const void* in_mask, dh_stream_t stream);
/* The same with a FIXED device footprint: resident_workgroups > 0 runs every slice as that many resident 512-thread workgroups
 * (two waves per SIMD, <= 80 registers) that walk the rows, so that the aggregation co-schedules with the 128 x 128 dh_gemm_f32_ex
 * of the next column panel instead of displacing its workgroups (the layer of scdsc.py:496-501 as one overlapped pipeline);
 * 0 = the one-shot grid of dh_spmm_csr_relu_slices_f32.  Bit-identical results.                                           */
DH_API int dh_spmm_csr_relu_slices_resident_f32(int64_t n_list, const int32_t* row_ids, int64_t n_cols, int64_t width,
                         int64_t slice_begin, int64_t slice_end,
                         const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* Z, int64_t ldz, float* Y, int64_t ldy, const float* bias, int act,
                         void* out_mask, const void* in_mask, int resident_workgroups, dh_stream_t stream);
/* Student-t soft assignment of the DEC clustering heads — q = normalise_j((1 / ((1 + ||z_i - mu_j||^2 / a) + eps))^pw * scale) — and its
 * backward (dZ may be NULL; dMU is always produced) without the [N, C, d] broadcast tensor the reference materialises:
 * SimpleGCDEC.forward (spagcn.py:391-397: a = alpha, eps = 1e-8, pw = alpha + 1, scale = 1/2), GC_DEC.forward (:600-608: eps = 1e-6),
 * ScDSCModel.forward (scdsc.py:466-468: a = v, eps = 0, pw = (v + 1) / 2, scale = 1).  c <= 64 clusters, c * d <= 4096 and
 * c d + 128 (c + d + 2) <= 16384 floats of LDS (dh_student_t_supported; e.g. 10 x 50, 20 x 60, 32 x 40).  G = d loss / d q.  Deterministic (block partials of dMU summed in a fixed order).                   */
DH_API int dh_student_t_supported(int64_t c, int64_t d);
DH_API int dh_student_t_forward_f32(int64_t n, int64_t c, int64_t d, const float* Z, int64_t ldz, const float* MU, float a, float eps,
                         float pw, float scale, float* Q, int64_t ldq, dh_stream_t stream);
DH_API size_t dh_student_t_backward_workspace_bytes(int64_t n, int64_t c, int64_t d);
DH_API int dh_student_t_backward_f32(int64_t n, int64_t c, int64_t d, const float* Z, int64_t ldz, const float* MU, float a, float eps,
                         float pw, float scale, const float* G, int64_t ldg, float* dZ, int64_t lddz, float* dMU,
                         void* workspace, size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_gather_rows_f32(int64_t n, int64_t width, const int32_t* idx, const float* X, int64_t ldx,
                       const void* relu_mask, float* out, int64_t ldo, dh_stream_t stream);
/* out = X * [Y > 0] from the recorded sign mask of Y (dh_relu_mask_bytes layout; width % 128 == 0): autograd's ReluBackward
 * (scdsc.py:499-500) as ONE streaming pass when the masked matrix is wanted as such — the default backward of the fused layer
 * masks dY once with this kernel and then runs the plain gather (4 requests per neighbour instead of 5).                 */
DH_API int dh_relu_mask_apply_f32(int64_t n_rows, int64_t width, const float* X, int64_t ldx, const void* relu_mask,
                       float* out, int64_t ldo, dh_stream_t stream);

/* ---- K2: deterministic CSR transpose (CSR of A^T) ------------------------------------------
 * out_perm[p] = index into the input nnz arrays of output entry p; output rows are ordered by
 * input position, i.e. exactly a stable sort by column (what scipy's tocsc gives).
 * val/out_val may both be NULL (pattern only).  workspace: dh_csr_transpose_workspace_bytes. */
DH_API size_t dh_csr_transpose_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz);
DH_API int dh_csr_transpose(int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int32_t* rowptr, const int32_t* col, const float* val,
                     int32_t* out_rowptr, int32_t* out_col, float* out_val, int32_t* out_perm,
                     void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* ---- K3: dense feature GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32) ---------------
 * C[M,N] (+)= op(A)[M,K] * op(B)[K,N]; exact f32 (one rounding per product, k-ordered chain).
 * trans_a=0: A is [M,K] (lda>=K); trans_a=1: A is stored [K,M] (lda>=M).  Same for B.
 * Replaces torch.mm(features, self.weight) (scdsc.py:497, spagcn.py:358) and its autograd
 * products dW = X^T dZ (trans_a=1, split over K) and dX = dZ W^T (trans_b=1).
 * accumulate != 0 adds into C.  workspace (split-K partial slabs): dh_gemm_f32_workspace_bytes;
 * may be NULL when that query returns 0.                                                       */
DH_API size_t dh_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b);
DH_API int dh_gemm_f32(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, int accumulate,
                void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* dh_gemm_f32 with an explicit macro-tile: 256x256 (one block of 8 wavefronts per CU, ~250 VGPRs: fewest bytes per flop) or
 * 128x128 (two blocks of 4 wavefronts per CU at ~170 VGPRs, which leaves a third of every SIMD's register file to the
 * wavefronts of an HBM-bound kernel running next to it on another stream: the pipelined layer's torch.mm, scdsc.py:497).
 * AUTO is what dh_gemm_f32 does.                                                                                   */
enum dh_gemm_tile { DH_GEMM_TILE_AUTO = 0, DH_GEMM_TILE_256 = 1, DH_GEMM_TILE_128 = 2 };
DH_API size_t dh_gemm_f32_ex_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, int tile);
DH_API int dh_gemm_f32_ex(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, int accumulate,
                void* workspace, size_t workspace_bytes, int tile, dh_stream_t stream);
/* The same product with torch.nn.Linear's bias (one value per column of C; NULL = none) and an optional ReLU applied in the store of
 * the output tile instead of a second pass over C (dance's Linear layers: scdsc.py:535-555 autoencoder, scdeepsort.py:80 classifier,
 * gnn.py:54 AdaptiveSAGE update).  Same arithmetic and rounding as dh_gemm_f32 followed by dh_bias_act_f32.                        */
DH_API int dh_gemm_f32_bias_act(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda,
                         const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                         void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* The same product (no accumulate) for the launch-bound shapes of a mini-batch step — K <= 512 and M N <= 2^20
 * (dh_gemm_f32_small_supported): graph-sc's WeightedGraphConv / Linear on a block of 128 cells (graphsc.py:452-467, :352-363) and
 * scDeepSort's classifier on a batch of 500 (scdeepsort.py:80-88).  32 x 32 tiles, the whole K extent fetched in one round trip, no
 * workspace.  Exact fp32 matrix-core arithmetic like dh_gemm_f32, in another summation order (results agree to rounding).          */
DH_API int dh_gemm_f32_small_supported(int64_t M, int64_t N, int64_t K);
DH_API int dh_gemm_f32_small(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda,
                      const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act, dh_stream_t stream);

/* Same contract as dh_gemm_f32 (operands, result and accumulation in fp32), computed on the bf16 matrix cores: each fp32
 * operand is split exactly into three bf16 terms and six of the nine partial products are accumulated in fp32 (the dropped
 * ones are <= 2^-23 of a product; csrc/gemm_f32x3.hip) — fp32-level accuracy at 6/16 of the fp32 matrix-pipe time.
 * Inputs must be finite.  Problems too small or unaligned for the split kernel run dh_gemm_f32 itself.               */
DH_API size_t dh_gemm_f32x3_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b);
DH_API int dh_gemm_f32x3(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, int accumulate, void* workspace,
                  size_t workspace_bytes, dh_stream_t stream);

/* out = a X + b Y (Y == NULL: out = a X), fp32, both products and the sum rounded separately: scDSC's mixing of the GCN and autoencoder
 * streams, `(1 - sigma) * h + sigma * tra` (scdsc.py:454-460), is three torch kernels and two temporaries per layer (28 B per element
 * instead of 12; 8.7 ms of a 190 ms epoch at 1M cells), its backward `(1 - sigma) * g`.                                              */
DH_API int dh_axpby_f32(int64_t n_rows, int64_t width, float a, const float* X, int64_t ldx, float b, const float* Y, int64_t ldy, float* out,
                 int64_t ldo, dh_stream_t stream);

/* ---- elementwise / reductions used by the layers' backward ---------------------------------
 * dh_relu_backward_f32: G = dY where Y > 0 else 0 (autograd of F.relu, scdsc.py:499-500).
 * dh_colsum_f32: out[j] = sum_i X[i,j] (bias gradient of spagcn.py:360-361); deterministic
 * two-pass reduction, workspace from dh_colsum_f32_workspace_bytes.                            */
DH_API int dh_relu_backward_f32(int64_t n_rows, int64_t width, const float* Y, int64_t ldy,
                         const float* dY, int64_t lddy, float* G, int64_t ldg, dh_stream_t stream);
/* X[i,:] = act(X[i,:] + bias) in place: bias of nn.Linear in AdaptiveSAGE (dance/models/nn/gnn.py:56)
 * and of GraphConvolution on a dense adjacency (spagcn.py:360-361).  bias may be NULL.            */
DH_API int dh_bias_act_f32(int64_t n_rows, int64_t width, float* X, int64_t ldx, const float* bias, int act,
                    dh_stream_t stream);
/* `nn.CrossEntropyLoss(reduction="sum")` of scDeepSort's training step (scdeepsort.py:185, :242) and its gradient in one pass:
 * loss[0] = sum_i (logsumexp(x_i) - x_{i, y_i}) over the rows with y_i != ignore_index (torch's default -100),
 * d_logits (may be NULL) = softmax(x_i) - onehot(y_i) (zero rows where ignored): the autograd backward is d_logits times the upstream
 * scalar.  labels: int64, as torch holds them.  Deterministic (block partials in the workspace, summed in block order).            */
DH_API size_t dh_softmax_xent_sum_workspace_bytes(int64_t n, int64_t n_classes);
DH_API int dh_softmax_xent_sum_f32(int64_t n, int64_t n_classes, const float* logits, int64_t ldx, const int64_t* labels,
                            int64_t ignore_index, float* loss, float* d_logits, int64_t ldd, void* workspace, size_t workspace_bytes,
                            dh_stream_t stream);
/* SpaGCN Gaussian kernel: e = exp(-d^2 / (2 l^2)) (spagcn.py:249-251 calculate_p, :807-809 calc_adj_exp),
 * f32 like numpy.  out (may be NULL) receives e; rowsum (may be NULL) receives sum_j e[i,j], so
 * calculate_p / search_l stream the N x N distance matrix without materialising the kernel.
 * Also used on the value array of a kNN-truncated CSR (n_rows = 1).                              */
DH_API int dh_gaussian_kernel_f32(int64_t n_rows, int64_t n_cols, const float* D, int64_t ldd, double l,
                           float* out, int64_t ldo, float* rowsum, dh_stream_t stream);
DH_API size_t dh_colsum_f32_workspace_bytes(int64_t n_rows, int64_t width);
DH_API int dh_colsum_f32(int64_t n_rows, int64_t width, const float* X, int64_t ldx, float* out,
                  void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* ---- K9/A14: dense pairwise distance ------------------------------------------------------
 * out[i,j] = dist(x_i, x_j), f32 [n,n].  Replaces the numba kernel pairwise_distance at
 * dance/utils/matrix.py:164-180 (euclidean :100-105: f32 differences and squares, accumulated in
 * double in index order, sqrt in double, rounded to f32 once; pearson :108-116, evaluated in
 * double; spearman :119-157 = pearson on mean-ranked rows: rank with dh_rank_rows_f32 first and
 * pass DH_METRIC_SPEARMAN).  Used by SpaGCNGraph / SpaGCNGraph2D
 * (dance/transforms/graph/spatial_graph.py:60,75).                                             */
DH_API int dh_pairwise_distance_f32(int64_t n, int64_t d, const float* X, int64_t ldx,
                             float* out, int64_t ldo, int metric, dh_stream_t stream);
/* out[r,c] = mean rank (ties averaged) of X[r,c] within row r (matrix.py:119-140).             */
DH_API int dh_rank_rows_f32(int64_t n, int64_t d, const float* X, int64_t ldx, float* out, int64_t ldo,
                     dh_stream_t stream);

/* ---- K8: exact brute-force kNN ------------------------------------------------------------
 * For each query row q in [q_begin, q_end) the k nearest rows of X (the query itself included,
 * as sklearn / scanpy do), ordered by (d2, index): ties go to the lower index.
 * d2 = sum_t rn(rn(x_t - y_t)^2), every op a separate f32 round-to-nearest, terms added in
 * feature order (no |x|^2 - 2xy expansion), so index lists are reproducible bit for bit;
 * out_dist = sqrt(d2).  Fewer than k points: remaining slots get index -1, distance +inf.
 * Replaces sklearn NearestNeighbors.kneighbors at dance/transforms/graph/heteronet_graph.py:36-37,
 * dance/transforms/graph/spatial_graph.py:147-149 and the kNN stage of sc.pp.neighbors
 * (dance/transforms/graph/neighbor_graph.py:52).  out_idx/out_dist are [(q_end-q_begin), k].
 * Two evaluation strategies with IDENTICAL results (the neighbours are defined by the sequential fp32 distance
 * chain and the (d2, index) order; both return exactly those):
 *   DH_KNN_SCAN   — every (query, candidate) pair through the chain on the vector ALUs (knn.hip);
 *   DH_KNN_FILTER — an upper bound of each query's k-th distance from a strided sample, a matrix-core pass that
 *                   discards every pair provably beyond it (d <= 64: one fp16 term on centred, power-of-two scaled
 *                   rows, the bound tightened in up to three passes over strided subsets of the candidates;
 *                   d > 64: three bf16 terms), and the chain on the survivors only (knn_filter.hip; k <= 64,
 *                   finite inputs);
 *   DH_KNN_AUTO   — FILTER for n >= 16384 candidates, >= 1024 queries and k <= 64, SCAN otherwise.
 * Workspace (64-byte aligned) from dh_knn_bruteforce_f32_workspace_bytes with the same algo.                     */
DH_API size_t dh_knn_bruteforce_f32_workspace_bytes(int64_t n, int64_t d, int64_t n_queries, int k, int algo);
DH_API int dh_knn_bruteforce_f32(int64_t n, int64_t d, const float* X, int64_t ldx,
                          int64_t q_begin, int64_t q_end, int k, int algo,
                          int32_t* out_idx, float* out_dist,
                          void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* Host only (no device): the plan DH_KNN_FILTER follows for d <= 64, for tests and capacity planning — same call site as
 * above (neighbor_graph.py:52).  out[0..n_out): 0 passes (0: d > 64, the tile kernel), 1 G (rows grouped by r % G), 2 H (the
 * first H slots hold the classes that are multiples of G / H), 3 slot reciprocal, 4 rows per class, 5 rows of the candidate
 * operand, 6 sample rows, 7 sample stride, 8 operand columns, 9 survivor slots per query, 10 candidate tiles, then per pass
 * p < 3 at 11 + 5 p: first tile, end tile, tiles per slice, segments, slots per segment.  Returns the number of fields (26). */
DH_API int dh_knn_filter_plan(int64_t n, int64_t d, int64_t n_queries, int k, int64_t* out, int n_out);

/* ---- K8: UMAP fuzzy-simplicial-set connectivities on a kNN list ------------------------------
 * What sc.pp.neighbors(method="umap") computes after its kNN search (neighbor_graph.py:52-55;
 * umap-learn smooth_knn_dist / compute_membership_strengths, local_connectivity = 1,
 * bandwidth = 1, 64 bisection steps, set_op_mix_ratio = 1).
 * dh_umap_membership_f32: rho_i = first positive distance, sigma_i by bisection (in double) so that
 *   sum_{j>=1} exp(-max(0,d_ij-rho_i)/sigma_i) = log2(k); w_ij = exp(-(d_ij-rho_i)/sigma_i), 1 where
 *   d_ij <= rho_i, 0 for the self slot / missing slots.  out_w is [n,k]; workspace >= 8 bytes.
 * dh_knn_row_nnz + dh_knn_graph_to_csr: drop w == 0 slots, sort each row by column -> CSR of W.
 * dh_csr_union_count + dh_csr_fuzzy_union_fill: sorted merge of W and W^T rows with
 *   v = (a + b) - a*b in f32 (W + W^T - W o W^T).  Row pointers come from dh_exclusive_scan_i32. */
DH_API int dh_umap_membership_f32(int64_t n, int k, const int32_t* knn_idx, const float* knn_dist,
                           float* out_w, float* out_sigma, float* out_rho,
                           void* workspace, size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_knn_row_nnz(int64_t n, int k, const int32_t* knn_idx, const float* w, int32_t* out_counts,
                   dh_stream_t stream);
DH_API int dh_knn_graph_to_csr(int64_t n, int k, const int32_t* knn_idx, const float* w,
                        const int32_t* rowptr, int32_t* out_col, float* out_val, dh_stream_t stream);
DH_API int dh_csr_union_count(int64_t n_rows, const int32_t* rowptr_a, const int32_t* col_a,
                       const int32_t* rowptr_b, const int32_t* col_b, int32_t* out_counts,
                       dh_stream_t stream);
DH_API int dh_csr_fuzzy_union_fill(int64_t n_rows, const int32_t* rowptr_a, const int32_t* col_a,
                            const float* val_a, const int32_t* rowptr_b, const int32_t* col_b,
                            const float* val_b, const int32_t* out_rowptr, int32_t* out_col,
                            float* out_val, dh_stream_t stream);
/* out[0] = 0, out[i+1] = in[0] + ... + in[i]  (out has n+1 entries).                            */
DH_API size_t dh_exclusive_scan_i32_workspace_bytes(int64_t n);
DH_API int dh_exclusive_scan_i32(int64_t n, const int32_t* in, int32_t* out, void* workspace,
                          size_t workspace_bytes, dh_stream_t stream);

/* ---- K10: CellFeatureGraph edge normalisation ----------------------------------------------
 * For every CSR row: out_val[e] = deg * val[e] / sum(val[row]) — the in-degree rescale of
 * dance/transforms/graph/cell_feature_graph.py:62-68 in one pass instead of a Python loop over
 * every node (row sums accumulated in double, rounded to f32 once).                            */
DH_API int dh_csr_row_normalize_f32(int64_t n_rows, const int32_t* rowptr, const float* val,
                             float* out_val, dh_stream_t stream);

/* Assemble the CellFeatureGraph of dance/transforms/graph/cell_feature_graph.py:38-69 as a CSR by
 * destination node from the expression matrix X (CSR, N cells x G genes) and its transpose
 * (dh_csr_transpose; perm_t = its out_perm).  Nodes: genes [0,G), cells [G,G+N).  Row of gene g: its
 * cell->gene in-edges then its self loop; row of cell c: its gene->cell in-edges then its self loop
 * (weight 1, :69).  out_eid[p] = the edge's id in the reference's edge order (cell->gene edges in
 * row-major nonzero order, then gene->cell, then self loops), so that order is recoverable exactly.
 * val_x / val_t are the (optionally dh_csr_row_normalize'd, :62-68) weights of X and X^T.
 * Outputs: out_rowptr [G+N+1], out_col/out_val/out_eid [2 nnz + G + N].                         */
DH_API int dh_cellgene_graph_assemble(int64_t n_cells, int64_t n_genes, int64_t nnz,
                               const int32_t* rowptr_x, const int32_t* col_x, const float* val_x,
                               const int32_t* rowptr_t, const int32_t* col_t, const float* val_t,
                               const int32_t* perm_t, int32_t* out_rowptr, int32_t* out_col,
                               float* out_val, int32_t* out_eid, dh_stream_t stream);

/* Dense expression matrix -> CSR in row-major non-zero order, i.e. ``row, col = np.nonzero(feat)`` of
 * dance/transforms/graph/cell_feature_graph.py:38 for a matrix that is already on the device (on-device preprocessing
 * pipeline): dh_dense_nnz_count_f32 gives the per-row counts, the caller scans them (dh_exclusive_scan_i32) and
 * dh_dense_to_csr_f32 fills col / val (ascending columns inside a row).                                             */
DH_API int dh_dense_nnz_count_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, int32_t* counts, dh_stream_t stream);
DH_API int dh_dense_to_csr_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const int32_t* rowptr,
                        int32_t* col, float* val, dh_stream_t stream);

/* ---- count-matrix normalisation feeding the graph builders (scanpy normalize_total / log1p / scale as the reference
 * pipelines call them: scdsc.py:113-131, sctag.py:119-139, transforms/normalize.py:531-679) --------------------------
 * dh_rowsum_masked_f32: out[r] = sum of X[r,c] over columns with colmask[c] != 0 (NULL = all), f64 accumulation;
 * dh_rowscale_log1p_f32: out = X / divisor[r] (divisor NULL = 1), then log1p(.) / ln(log_base) if do_log1p (log_base <= 0: natural);
 * dh_col_standardize_f32: out = (X - mean[c]) / std[c] with f64 statistics and f32 stores after each step (numpy's in-place
 *   arithmetic), clipped to [-max_value, max_value] (mean NULL: no centring and only the upper clip; max_value <= 0: none);
 * dh_col_moments_f32: partial[b][0][c] / partial[b][1][c] = f64 sums of X[r,c] / fl32(X[r,c]^2) over row block b
 *   (b < ceil(n_rows / rows_per_block) <= 65535), the caller adds the blocks;
 * dh_col_any_gt_f32: flag[c] = 1 iff X[r,c] > thresh[r] for some row r.
 * out may alias X.                                                                                                  */
DH_API int dh_rowsum_masked_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const uint8_t* colmask,
                         float* out, dh_stream_t stream);
DH_API int dh_rowscale_log1p_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const float* divisor,
                          int do_log1p, double log_base, float* out, int64_t ldo, dh_stream_t stream);
DH_API int dh_col_standardize_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const double* mean,
                           const double* std, double max_value, float* out, int64_t ldo, dh_stream_t stream);
DH_API int dh_col_moments_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, int64_t rows_per_block,
                       double* partial, dh_stream_t stream);
DH_API int dh_col_any_gt_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const float* thresh,
                      uint8_t* flag, dh_stream_t stream);

/* ---- kNN-truncated Gaussian spatial adjacency (SpaGCN at scale; spagcn.py:249-251,807-809 evaluate the DENSE kernel) ----
 * Rows = spots; each row keeps the spot's k nearest spots (exact, self included) with value exp(-d^2 / (2 l^2)) in the
 * reference's fp32 expression, columns ascending; l <= 0 writes the distances d instead.  out_rowptr [n+1],
 * out_col / out_val [n*k]; k <= min(n, 64).                                                                        */
DH_API size_t dh_spatial_gaussian_knn_workspace_bytes(int64_t n, int64_t d, int k);
DH_API int dh_spatial_gaussian_knn(int64_t n, int64_t d, const float* X, int64_t ldx, int k, double l,
                            int32_t* out_rowptr, int32_t* out_col, float* out_val,
                            void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* ---- edge softmax of graph attention (stagate.py:31-128 GATConv.message; scgnn2.py:1091-1118) --------------------
 * att[e] = softmax over the in-edges e = (j -> i) of row i of act(a_src[j] + a_dst[i]); act 0 = sigmoid (STAGATE),
 * 1 = leaky_relu(negative_slope) (GAT); denominator + 1e-16 as torch_geometric.utils.softmax.  a_dst may be NULL.
 * The weighted aggregation itself is dh_spmm_csr_f32 with val = att (no [E, C] message tensor).
 * Backward: dt[e] = act'(t_e) att[e] (datt[e] - sum_k att[k] datt[k]) per row, datt from dh_sddmm_csr_f32;
 * d_a_dst[i] = sum_e dt[e] (may be NULL); d a_src is the column-wise sum of dt (caller: scatter-add over col).      */
DH_API int dh_edge_softmax_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* col, const float* a_src,
                        const float* a_dst, int act, float negative_slope, float* att, dh_stream_t stream);
/* Same with the exponent shifted by *shift (a device scalar) instead of each row's own maximum: scGNN2's GATLayer subtracts the
 * GLOBAL maximum of all edge scores before exp (scgnn2.py:1071-1085), which with the + 1e-16 denominator is not the same
 * function as the row-shifted softmax.  shift == NULL: identical to dh_edge_softmax_f32.                                  */
DH_API int dh_edge_softmax_shift_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* col, const float* a_src,
                        const float* a_dst, int act, float negative_slope, const float* shift, float* att, dh_stream_t stream);
DH_API int dh_edge_softmax_backward_f32(int64_t n_rows, const int32_t* rowptr, const int32_t* col,
                                 const float* a_src, const float* a_dst, int act, float negative_slope,
                                 const float* att, const float* datt, float* dt, float* d_a_dst,
                                 dh_stream_t stream);

/* ---- two-hop adjacency pattern of scHeteroNet (scheteronet.py:507-539, HeteroNet.init_adj) ------------------
 * Pattern of ((A A) - A) > 0 for a 0/1 CSR pattern A (ascending, duplicate-free columns): (i, c) is kept when c is reached
 * from i by two edges and is not "used up" by a direct edge, i.e. (i, c) is not in A or has >= 2 two-edge paths; drop_diag
 * additionally removes c == i.  Symbolic SpGEMM on the device: path keys (i << 32 | c), radix sort, run heads.
 *   dh_csr_two_hop_count  : rowcnt[i] = number of two-edge paths starting at row i (*overflow = 1 if one row exceeds 2^31)
 *   dh_csr_two_hop_expand : offs = exclusive scan of rowcnt, total = offs[n]; writes flags[total] (1 at kept run heads)
 *   dh_csr_two_hop_compact: pos = exclusive scan of flags; out_rowptr [n+1], out_col [pos[total]]
 * The workspace (dh_csr_two_hop_workspace_bytes, 256-byte aligned) carries the sorted keys from expand to compact.   */
DH_API int dh_csr_two_hop_count(int64_t n, const int32_t* rowptr, const int32_t* col, int32_t* rowcnt,
                         int32_t* overflow, dh_stream_t stream);
DH_API size_t dh_csr_two_hop_workspace_bytes(int64_t n, int64_t total);
DH_API int dh_csr_two_hop_expand(int64_t n, int64_t total, const int32_t* rowptr, const int32_t* col,
                          const int32_t* offs, int drop_diag, int32_t* flags, void* workspace,
                          size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_csr_two_hop_compact(int64_t n, int64_t total, const int32_t* flags, const int32_t* pos,
                           int32_t* out_rowptr, int32_t* out_col, const void* workspace, dh_stream_t stream);

/* ---- dense part of graph-sc's inner-product decoder loss (graphsc.py:208-216) ------------------------------
 * binary_cross_entropy_with_logits(X, adj, pos_weight) with a target that is zero almost everywhere: element loss
 * softplus(x), derivative sigmoid(x); the y = 1 corrections are sparse and stay with the caller.
 * dh_softplus_rowsum_f32: rowsum[r] = sum_c softplus(X[r,c]) (f64 accumulation, deterministic);
 * dh_sigmoid_scale_f32:   out = scale[0] * sigmoid(X)  (scale is a DEVICE scalar: no host round trip).            */
DH_API int dh_softplus_rowsum_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, float* rowsum,
                           dh_stream_t stream);
DH_API int dh_sigmoid_scale_f32(int64_t n_rows, int64_t n_cols, const float* X, int64_t ldx, const float* scale,
                         float* out, int64_t ldo, dh_stream_t stream);

/* ---- the same decoder without the B x B logit matrix (gram_bce.hip) --------------------------------------------
 * Replaces, for GraphSC.fit (graphsc.py:208-216) with InnerProductDecoder (:405-411, adj_logits = z z^T): the logits
 * GEMM, the two passes above and the two B x B x d GEMMs of the backward.  One pass on the fp32 matrix cores:
 *   rowloss[i] = sum_j softplus(<z_i, z_j>),   O[i, :] = sum_j sigmoid(<z_i, z_j>) z_j     (i, j < n; Z [n, d] fp32)
 * so that  sum_ij softplus(x_ij) = sum_i rowloss[i]  and  d/dz of it = 2 O  (x = z z^T is symmetric).  d <= 320
 * (dh_gram_sigmoid_supported); fixed summation order, no atomics.  workspace: dh_gram_sigmoid_workspace_bytes, 16-byte
 * aligned.                                                                                                          */
DH_API int dh_gram_sigmoid_supported(int64_t n, int64_t d);
DH_API size_t dh_gram_sigmoid_workspace_bytes(int64_t n, int64_t d);
DH_API int dh_gram_sigmoid_f32(int64_t n, int64_t d, const float* Z, int64_t ldz, float* O, int64_t ldo, float* rowloss,
                        void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* The same pass for another element function f of the logits (rowloss[i] = sum_j f(x_ij), O[i] = sum_j f'(x_ij) z_j):
 * DH_GRAM_SOFTPLUS = dh_gram_sigmoid_f32; DH_GRAM_SIGMOID_SQ: f = sigmoid^2 — the dense part sum_ij sigmoid(x_ij)^2 of scTAG's
 * adjacency reconstruction loss mean((sigmoid(z0 z0^T) - adj)^2) (sctag.py:470-471, :254), whose sparse part
 * (-2 a_ij sigmoid(x_ij) + a_ij^2 on the edges) is an SDDMM (dh_sddmm_csr_f32): the N x N matrix never exists.          */
enum dh_gram_mode { DH_GRAM_SOFTPLUS = 0, DH_GRAM_SIGMOID_SQ = 1 };
DH_API int dh_gram_pairwise_f32(int mode, int64_t n, int64_t d, const float* Z, int64_t ldz, float* O, int64_t ldo, float* rowloss,
                         void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* The rectangular form of the same pass: rows i from Zr [n_rows, d], columns j from Z [n, d] —
 *   rowloss[i] = sum_{j < n} f(<zr_i, z_j>),   O[i, :] = sum_{j < n} f'(<zr_i, z_j>) z_j        (i < n_rows)
 * — for the cell-sharded full-graph decoder (dance_amd/sharding.py: a rank's rows of graphsc.py:208-216's loss against the
 * all-gathered embedding; with x symmetric the gradient of the GLOBAL sum w.r.t. a rank's own rows is still 2 O, so no gradient
 * travels).  dh_gram_pairwise_f32 is this with Zr = Z.  workspace: dh_gram_pairwise_rect_workspace_bytes.                  */
DH_API size_t dh_gram_pairwise_rect_workspace_bytes(int64_t n_rows, int64_t n, int64_t d);
DH_API int dh_gram_pairwise_rect_f32(int mode, int64_t n_rows, int64_t n, int64_t d, const float* Zr, int64_t ldr, const float* Z,
                              int64_t ldz, float* O, int64_t ldo, float* rowloss, void* workspace, size_t workspace_bytes,
                              dh_stream_t stream);

/* The y = 1 entries of the same loss (graphsc.py:208-214: the non-zeros of adj[dst][:, dst]), listed as (us[e], vs[e]), each at
 * most once: forward xe[e] = <z_us, z_vs>, term[e] = pos_weight * softplus(-xe) - softplus(xe) (add sum(term) to sum(rowloss));
 * backward dZ = scale[0] * (2 O + sum_e c_e (e_us z_vs^T + e_vs z_us^T)), c_e = pos_weight (sigmoid(xe) - 1) - sigmoid(xe), in list
 * order (no atomics); scale is a DEVICE scalar.  Written in round 2, not yet on GraphSC.fit's default path.              */
DH_API int dh_gram_listed_forward_f32(int64_t n, int64_t d, int64_t n_listed, const float* Z, int64_t ldz, const int32_t* us,
                               const int32_t* vs, float pos_weight, float* xe, float* term, dh_stream_t stream);
DH_API int dh_gram_listed_backward_f32(int64_t n, int64_t d, int64_t n_listed, const float* Z, int64_t ldz, const float* O,
                                int64_t ldo, const int32_t* us, const int32_t* vs, const float* xe, float pos_weight,
                                const float* scale, float* dZ, int64_t ldd, dh_stream_t stream);
/* the same backward when the listed entries are exactly (i, i) for i < n (the identity target of a batch of cells of a cell - gene graph:
 * its only cell -> cell edges are the self loops, graphsc.py:208-214), xe[i] = <z_i, z_i>: one elementwise pass, same arithmetic order. */
DH_API int dh_gram_diag_backward_f32(int64_t n, int64_t d, const float* Z, int64_t ldz, const float* O, int64_t ldo, const float* xe, float pos_weight,
                              const float* scale, float* dZ, int64_t ldd, dh_stream_t stream);

/* ---- message-flow blocks of the full-neighbour sampler (block.hip) -------------------------------------------
 * What dgl.dataloading.NeighborSampler([-1]*L, edge_dir="in") / MultiLayerFullNeighborSampler produce for a batch of
 * seed nodes (scdeepsort.py:183,233-236; graphsc.py:181-183): every in-edge of the seeds; source nodes = the seeds first,
 * then the remaining in-neighbours by ascending id; block columns index that source list.
 *   dh_block_plan: block_rowptr [n_seeds+1] (scan of the seed degrees) and totals[2] = {block edges, non-seed sources};
 *                  the caller reads `totals` (its only host round trip) to size the outputs of
 *   dh_block_fill: src_ids int64 [n_seeds + totals[1]], block_col int32 / block_val f32 [totals[0]] (block_val may be NULL
 *                  together with val).
 * mark: uint8 [n_nodes], all zero on entry and again on return of dh_block_fill; lut: int32 [n_nodes] scratch (node ->
 * position in src_ids, valid for the block's sources only).  Both persist with the graph; workspace (same buffer for
 * both calls) from dh_block_workspace_bytes.  Seeds must be unique.                                              */
DH_API size_t dh_block_workspace_bytes(int64_t n_nodes, int64_t n_seeds);
DH_API int dh_block_plan(int64_t n_nodes, int64_t n_seeds, const int64_t* seeds, const int32_t* rowptr,
                  const int32_t* col, uint8_t* mark, int32_t* lut, int32_t* block_rowptr, int32_t* totals,
                  void* workspace, size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_block_fill(int64_t n_nodes, int64_t n_seeds, const int64_t* seeds, const int32_t* rowptr,
                  const int32_t* col, const float* val, uint8_t* mark, int32_t* lut,
                  const int32_t* block_rowptr, int32_t* block_col, float* block_val, int64_t* src_ids,
                  void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* Static-shape block of seed CELLS of a CellFeatureGraph-layout graph (genes = nodes [0, n_genes); a cell's in-edges are its
 * genes, ascending, then its self loop): sources = [the B seeds | all genes] — a superset of dgl.to_block's source list that gives
 * the same layer outputs — and exactly e_max stored entries: rows 0..B-1 the seeds' in-edges (column = B + gene, or the row's own
 * index for the self loop), row B a padding row over the unused tail (column 0, value 0).  block_rowptr has B + 2 entries.  No
 * value is read back by the host, every shape is static: a training step over such a block (scdeepsort.py:183-262,
 * graphsc.py:181-218 at the reference's batch sizes) can be captured as one hipGraph.  bad[0] (int32, device) collects flags:
 * bit 1 = a seed does not have that layout or the block needs more than e_max entries, bit 2 = a seed without exactly one self
 * loop (graph-sc's identity decoder target needs one; scDeepSort's step does not).                                             */
DH_API size_t dh_block_cells_static_workspace_bytes(int64_t n_seeds);
DH_API int dh_block_cells_static(int64_t n_seeds, int64_t n_genes, int64_t e_max, const int64_t* seeds, const int32_t* rowptr,
                          const int32_t* col, const float* val, int32_t* block_rowptr, int32_t* block_col, float* block_val,
                          int32_t* bad, void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* Degree scalings of a block for dgl.nn.GraphConv's norms as graph-sc's WeightedGraphConv applies them (dance graphsc.py:444-474):
 * DH_DEGREE_BOTH: rowscale[i] = max(indeg(i), 1)^-1/2, colscale[j] = max(outdeg(j), 1)^-1/2; DH_DEGREE_MEAN: rowscale[i] = 1 / max(indeg(i), 1)
 * (colscale, col, count unused).  Degrees are counted over the entries of rows [0, n_rows) — rowptr has at least n_rows + 1 entries; a
 * static block's padding row and tail lie behind them; rowscale has n_rows + n_pad entries, the last n_pad (<= 256) are set to 1.
 * count: caller-owned int32 [n_cols] scratch.  Three stream operations, no host read. */
enum { DH_DEGREE_BOTH = 0, DH_DEGREE_MEAN = 1 };
DH_API int dh_csr_degree_scales_f32(int64_t n_rows, int64_t n_pad, int64_t n_cols, const int32_t* rowptr, const int32_t* col, int mode,
                             float* rowscale, float* colscale, int32_t* count, dh_stream_t stream);

/* ---- K4/K7: AdaptiveSAGE message + mean aggregation -----------------------------------------
 * neigh[v,:] = mean_{e=(u->v)} alpha[idx(e)] * w_e * H[u,:], idx(e) chosen from the src/dst
 * "cell_id" arrays exactly as dance/models/nn/gnn.py:72-76 (gene->cell: src id; cell->gene:
 * dst id; gene self loop: G; cell self loop: G+1); mean over in-edges, 0 for an isolated node
 * (DGL fn.mean, gnn.py:90).  CSR rows = dst nodes of the block, columns = src nodes.
 * dh_sage_alpha_grad_f32: dalpha[idx(e)] += w_e * <H[u], dneigh[v]> / deg(v)  (K7; zeroed first;
 * float atomics: summation order, hence the last bits, is not reproducible).                   */
DH_API int dh_sage_aggregate_f32(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes,
                          const int32_t* rowptr, const int32_t* col, const float* w,
                          const int32_t* src_cell_id, const int32_t* dst_cell_id,
                          const float* alpha, const float* H, int64_t ldh,
                          float* neigh, int64_t ldn, dh_stream_t stream);
/* Densified-operand form of the same aggregation (densify.hip): at 10 % density the MFMA GEMM over a dense copy of the
 * weighted adjacency beats every vector-ALU gather (measurements in DESIGN.md), most of all for the gene <- cell rows of
 * ~1e5 in-edges.  dh_csr_densify_window writes out[r][c - col_begin] = val_e * rowscale[r] * colscale[c - col_begin]
 * (* 1/deg(r) if mean) for the edges of row r with col_begin <= c < col_begin + n_cols and 0 elsewhere (fp32 or bf16);
 * max_row_nnz (an upper bound of the longest row) only sizes the launch of the wide-window path (n_cols > 16384, at most
 * 65535 rows).  dh_sage_tail computes the mean-scaled contribution of the edges whose source lies OUTSIDE that window
 * (the self loops) with the alpha rule of dh_sage_aggregate_f32; the host then accumulates the GEMM into it:
 *     neigh = dh_sage_tail(...);  neigh += A_dense * H[col_begin : col_begin + n_cols]   (dh_gemm_f32 / dh_gemm_bf16).  */
DH_API int dh_csr_densify_window(int64_t n_rows, int64_t max_row_nnz, const int32_t* rowptr, const int32_t* col,
                          const float* val, const float* rowscale, const float* colscale, int mean,
                          int64_t col_begin, int64_t n_cols, void* out, int64_t ldo, int out_dtype,
                          dh_stream_t stream);
DH_API int dh_sage_tail(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes, int64_t col_begin, int64_t n_cols,
                 const int32_t* rowptr, const int32_t* col, const float* w, const int32_t* src_cell_id,
                 const int32_t* dst_cell_id, const float* alpha, const void* H, int64_t ldh, int h_dtype,
                 void* neigh, int64_t ldn, int out_dtype, dh_stream_t stream);
/* The same window product on the matrix cores WITHOUT the dense copy: neigh[v,:] += 1/deg(v) * sum over the in-edges
 * (u -> v) with col_begin <= u < col_begin + n_cols of (w_e * colscale[u - col_begin]) * H[u,:].  A workgroup densifies its
 * 128 cells x 128 window columns at a time into LDS in MFMA fragment order (adjacency entries and fp32 features split
 * into bf16 hi + lo, fp32 accumulation: ~1e-5 worst-case relative error per term).  With src_cell_id / dst_cell_id / alpha
 * (all three or none) the out-of-window in-edges (the self loops, at the rows' ends) are added in the epilogue with the
 * alpha rule of dh_sage_aggregate_f32 and neigh is WRITTEN (the whole AdaptiveSAGE mean in one launch); without them neigh
 * is accumulated into (dh_sage_tail initialises it).  Preconditions: inside a row the in-window edges are contiguous and ascending by column
 * (CellFeatureGraph / block layout), width <= 448, n_cols <= 4096.  nnz = length of col / w (bounds the stream prefetch);
 * workspace: dh_sage_window_mfma_workspace_bytes (the K-permuted bf16 planes of the window's feature rows).           */
DH_API int dh_sage_window_mfma_supported(int64_t n_cols, int64_t width, int h_dtype);  /* 1 if the shape fits the kernel's LDS plan */
DH_API size_t dh_sage_window_mfma_workspace_bytes(int64_t n_cols, int64_t width, int h_dtype);
/* Workspace that also holds the fp32 shares of a SPLIT launch: with this many bytes a launch of few destination rows (a mini-batch: fewer
 * than 256 blocks of 128 rows) splits the gene window over up to 512 / blocks workgroups per row block and sums the shares in a second,
 * deterministic kernel (scDeepSort's batch of 500 cells: 217 -> ~30 us); with the smaller size above the unsplit kernel runs. */
DH_API size_t dh_sage_window_mfma_split_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype);
/* Workspace of the round-4 kernel pair behind UNSPLIT launches (>= 256 blocks of 128 rows: the full graph, large batches) with the
 * self loops folded in: a repack of the rows into per-64-row, K-chunk-major entry lists with the entries already scaled and split
 * (8 bytes per stored entry of col / w: nnz of them), the chunk pointers and the feature planes; then a two-wavefronts-per-SIMD MFMA
 * kernel whose loop moves one packed entry per lane and chunk.  Give dh_sage_window_mfma at least this many bytes and it takes that
 * path (same result to fp32 rounding: the K order of the sums differs); with less it runs the single-wave kernel.  0 = shape not
 * supported (width > 512, n_cols > 4096).  Also needs 16-byte aligned feature rows (H, ldh * size, width * size multiples of 16). */
DH_API size_t dh_sage_window_mfma_bcm_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype, int64_t nnz);
DH_API int dh_sage_window_mfma(int64_t n_dst, int64_t n_src, int64_t width, int64_t col_begin, int64_t n_cols,
                        const int32_t* rowptr, const int32_t* col, const float* w, const float* colscale,
                        const void* H, int64_t ldh, int h_dtype, void* neigh, int64_t ldn, int out_dtype, int64_t nnz,
                        const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha, int64_t n_genes,
                        void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* The plan interface of the same kernel pair, for callers that aggregate over ONE graph many times (ScDeepSort.fit evaluates the full
 * graph three times per epoch, scdeepsort.py:191-193; alpha changes between calls, the graph does not).  dh_sage_window_plan repacks
 * the rows [0, n_dst) once — per 64-row group the in-window entries sorted by 32-gene K chunk, 8 bytes each, plus the chunk pointers:
 * it depends on rowptr / col / w only.  dh_sage_window_mfma_planned then computes what dh_sage_window_mfma computes (self loops folded
 * in: src_cell_id / dst_cell_id / alpha required) with workspace = dh_sage_window_mfma_planned_workspace_bytes (the feature planes).
 * Supported shapes: dh_sage_window_mfma_planned_supported (width <= 512, n_cols <= 4096, 16-byte aligned feature rows). */
DH_API size_t dh_sage_window_plan_bytes(int64_t n_dst, int64_t n_cols, int64_t nnz);
DH_API int dh_sage_window_plan(int64_t n_dst, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                        const float* w, int64_t nnz, void* plan, size_t plan_bytes, dh_stream_t stream);
DH_API int dh_sage_window_mfma_planned_supported(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype, const void* H, int64_t ldh,
                                          int64_t nnz);
DH_API size_t dh_sage_window_mfma_planned_workspace_bytes(int64_t n_cols, int64_t width, int h_dtype);
DH_API int dh_sage_window_mfma_planned(int64_t n_dst, int64_t n_src, int64_t width, int64_t col_begin, int64_t n_cols,
                                const int32_t* rowptr, const int32_t* col, const float* w, const float* colscale, const void* H,
                                int64_t ldh, int h_dtype, void* neigh, int64_t ldn, int out_dtype, int64_t nnz,
                                const int32_t* src_cell_id, const int32_t* dst_cell_id, const float* alpha, int64_t n_genes,
                                const void* plan, size_t plan_bytes, void* workspace, size_t workspace_bytes, dh_stream_t stream);
/* The other direction of the same layer — GENE destinations (gnn.py:74: a cell -> gene edge carries alpha[cell_id of the destination
 * gene]; a 2-layer scDeepSort, scdeepsort.py:183 `[-1] * L`, aggregates into its gene nodes every pass): few rows (n_dst <= 65536; 2000
 * genes), each with ~1e5 in-edges from a window [col_begin, col_begin + n_cols) of up to 2^30 source rows (every cell).  The window
 * is the K dimension of the matrix-core loop of dh_sage_window_mfma_planned, cut into slices of 2048 columns that are dealt to up to
 * 256 workgroups (a set of slices per workgroup, the workgroups of one set on one XCD so that they share its L2 for the feature
 * blocks); the fp32 shares are summed in set order by a second kernel (deterministic), which also applies
 *     neigh[v,:] = 1/deg(v) * ( rowscale[v] * sum_{in-window e=(u->v)} w_e H[u,:]  +  sum_{other e} alpha[idx(e)] w_e H[u,:] )
 * with idx(e) the rule of dh_sage_aggregate_f32 for the out-of-window in-edges (the self loops).  rowscale may be NULL (= 1).
 * No dense copy of the adjacency (2000 x 1e6 fp32 = 8 GB) is ever formed.  fp32 features and the weights enter as bf16 hi + lo pairs
 * (three products per term, fp32 accumulation: ~1e-5 relative per term); bf16 features exactly.
 * dh_sage_window_splitk_plan repacks the in-window entries once per graph (8 bytes per stored entry + pointer tables, depends on
 * rowptr / col / w only); workspace = the feature planes of the window + the shares.  Precondition as above: inside a row the
 * in-window edges are contiguous and ascending by column. */
DH_API size_t dh_sage_window_splitk_plan_bytes(int64_t n_dst, int64_t n_cols, int64_t nnz);
DH_API int dh_sage_window_splitk_plan(int64_t n_dst, int64_t col_begin, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                               const float* w, int64_t nnz, void* plan, size_t plan_bytes, dh_stream_t stream);
DH_API int dh_sage_window_splitk_supported(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype, const void* H, int64_t ldh,
                                    int64_t nnz);
DH_API size_t dh_sage_window_splitk_workspace_bytes(int64_t n_dst, int64_t n_cols, int64_t width, int h_dtype);
DH_API int dh_sage_window_splitk(int64_t n_dst, int64_t n_src, int64_t width, int64_t col_begin, int64_t n_cols, const int32_t* rowptr,
                          const int32_t* col, const float* w, const float* rowscale, const void* H, int64_t ldh, int h_dtype,
                          void* neigh, int64_t ldn, int out_dtype, int64_t nnz, const int32_t* src_cell_id,
                          const int32_t* dst_cell_id, const float* alpha, int64_t n_genes, const void* plan, size_t plan_bytes,
                          void* workspace, size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_sage_alpha_grad_f32(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes,
                           const int32_t* rowptr, const int32_t* col, const float* w,
                           const int32_t* src_cell_id, const int32_t* dst_cell_id,
                           const float* H, int64_t ldh, const float* dneigh, int64_t ldn,
                           float* dalpha, dh_stream_t stream);

/* ---- SDDMM ------------------------------------------------------------------------------------------------
 * out[e] = scale[e] * <U[row(e), :], V[col(e), :]> for every stored edge e of a CSR pattern (scale may be NULL).
 * The edge-value gradient of dh_spmm_csr_f32 (dval[e] = <dY[row(e)], Z[col(e)]>: backward of the edge-weighted
 * aggregations of graphsc.py:417-426 and gnn.py:81-82 with respect to the weights) and the per-edge logits of the
 * inner-product decoder graphsc.py:386-411 restricted to stored edges.  width % 4 == 0, 16-byte aligned rows,
 * width <= 2048.                                                                                              */
DH_API int dh_sddmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t width,
                     const int32_t* rowptr, const int32_t* col, const float* scale,
                     const float* U, int64_t ldu, const float* V, int64_t ldv,
                     float* out, dh_stream_t stream);
/* the same with bf16-stored U / V (fp32 products, sums and result) */
DH_API int dh_sddmm_csr_bf16(int64_t n_rows, int64_t n_cols, int64_t width,
                      const int32_t* rowptr, const int32_t* col, const float* scale,
                      const uint16_t* U, int64_t ldu, const uint16_t* V, int64_t ldv,
                      float* out, dh_stream_t stream);

/* ---- bf16 storage path (scDeepSort at 1M cells, "bf16 with MFMA dense update") --------
 * Features, activations and their gradients are STORED as bf16 (uint16_t bit patterns, torch.bfloat16); every sum is
 * accumulated in fp32 and rounded once (nearest-even) when the output dtype is DH_DTYPE_BF16.  The reference has no
 * reduced-precision path (it is fp32 DGL/torch on CPU): parity is <= 1e-2 relative against the fp32 oracle
 * evaluated on the same bf16-rounded inputs, and the fp32 entry points above remain the default.
 * Rows must be multiples of 8 features and 16-byte aligned (one lane moves 8 bf16 per load).
 *
 * dh_spmm_csr_bf16 / dh_sage_aggregate_bf16: dh_spmm_csr_f32 / dh_sage_aggregate_f32 with a bf16 gathered operand —
 *   the aggregation of gnn.py:62-90 and graphsc.py:462-465 at half the HBM bytes per neighbour row.
 * dh_gemm_bf16: C = act(op(A) op(B) + bias) (+ C), bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16 —
 *   nn.Linear of AdaptiveSAGE (gnn.py:56-58,93-94: x W^T + b, then the activation) and its two gradient products.
 *   op(A) is [M,K] (stored [K,M] if trans_a), op(B) is [K,N] (stored [N,K] if trans_b); bias [N] fp32 or NULL.
 *   K-contiguous operands are read in place when their rows are 16-byte aligned (base pointer and leading dimension
 *   multiples of 8 elements, K % 8 == 0); the workspace query assumes that, anything else is repacked into the
 *   workspace (K-strided operands always are) — except the weight gradient dW = dY^T X of a Linear layer (trans_a = 1, trans_b = 0,
 *   M <= 224, M and N multiples of 8, K >= 4096, 16-byte aligned rows): both operands are then streamed as they lie and
 *   transposed in registers on their way into LDS, no transposed copies in HBM (1.61 -> 0.41 ms at 200 x 400 x 1M).
 * dh_relu_backward_bf16 / dh_colsum_bf16: G = dY * [Y > 0]; out[j] = sum_i X[i,j] in fp32 (workspace as
 *   dh_colsum_f32_workspace_bytes).                                                                              */
DH_API int dh_spmm_csr_bf16(int64_t n_rows, int64_t n_cols, int64_t width,
                     const int32_t* rowptr, const int32_t* col, const float* val,
                     const float* rowscale, const float* colscale,
                     const uint16_t* Z, int64_t ldz, void* Y, int64_t ldy, int y_dtype,
                     const float* bias, int act, int reduce, dh_stream_t stream);
DH_API int dh_sage_aggregate_bf16(int64_t n_dst, int64_t n_src, int64_t width, int64_t n_genes,
                           const int32_t* rowptr, const int32_t* col, const float* w,
                           const int32_t* src_cell_id, const int32_t* dst_cell_id,
                           const float* alpha, const uint16_t* H, int64_t ldh,
                           void* neigh, int64_t ldn, int neigh_dtype, dh_stream_t stream);
DH_API size_t dh_gemm_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b);
DH_API int dh_gemm_bf16(int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                 const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                 void* C, int64_t ldc, int c_dtype, const float* bias, int act, int accumulate,
                 void* workspace, size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_relu_backward_bf16(int64_t n_rows, int64_t width, const uint16_t* Y, int64_t ldy,
                          const uint16_t* dY, int64_t lddy, uint16_t* G, int64_t ldg, dh_stream_t stream);
DH_API int dh_colsum_bf16(int64_t n_rows, int64_t width, const uint16_t* X, int64_t ldx, float* out,
                   void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* ---- the narrow GCN layer fused (gcn_narrow.hip): in < 64, out <= 64 -----------------------------------------------------
 * SpaGCN's GraphConvolution 50 -> 50 (spagcn.py:357-363: spmm(adj, mm(input, weight)) + bias) and any layer of that size,
 * evaluated aggregate-first — (A X) W + b, the same function as A (X W) + b — so that the forward is ONE gather kernel and the
 * weight / bias gradient ONE streaming kernel on the fp32 matrix cores (instead of skinny GEMM + SpMM, resp. colsum + SpMM +
 * split-K GEMM).  forward: Y = act((A X) W + bias); if agg != NULL it receives [n_rows, 64] fp32 (16-byte aligned): the
 * aggregated rows, zero padded, column 63 = 1 — the operand of the backward.  X rows 8-byte aligned (16-byte: four rows per
 * wavefront instead of two).  backward: dW [in, out] = agg^T G, db [out] = 1^T G (db may be NULL) with G = dY, or
 * dY * [Y_act > 0] when Y_act (the layer's ReLU output) is given; deterministic.  workspace: dh_gcn_narrow_backward_workspace_bytes.
 * The input gradient A^T (G W^T), when needed, is dh_gemm_f32 + dh_spmm_csr_f32.                                          */
DH_API int dh_gcn_narrow_supported(int64_t in_features, int64_t out_features);
DH_API int dh_gcn_narrow_forward_f32(int64_t n_rows, int64_t n_cols, int64_t in_features, int64_t out_features,
                              const int32_t* rowptr, const int32_t* col, const float* val, const float* X, int64_t ldx,
                              const float* W, int64_t ldw, const float* bias, int act, float* agg, float* Y, int64_t ldy,
                              dh_stream_t stream);
DH_API size_t dh_gcn_narrow_backward_workspace_bytes(int64_t n_rows);
DH_API int dh_gcn_narrow_backward_f32(int64_t n_rows, int64_t in_features, int64_t out_features, const float* agg,
                               const float* dY, int64_t ldd, const float* Y_act, int64_t ldy, float* dW, int64_t ldw, float* db,
                               void* workspace, size_t workspace_bytes, dh_stream_t stream);

/* ---- fused zero-inflated negative-binomial NLL (zinb.hip) ---------------------------------------------------------------
 * ZINBLoss.forward of dance/utils/loss.py:780-829 (scTAG sctag.py:254,347; scDSC scdsc.py:279-283; scHeteroNet
 * scheteronet.py:289-336): ~25 elementwise torch passes over four N x G matrices, in float64 after the size factors promote
 * the expression.  forward: one read of X (raw counts), mean, disp, pi (fp32, row-major) -> rowloss[i] = sum_g loss(i, g) in
 * float64 (the caller takes sum / (n g)); backward: recomputes the element terms and writes d mean, d disp, d pi (fp32,
 * leading dimension ldo) times *upstream (a DEVICE float64 scalar = grad_output / (n g)).  scale_factor: float64 [n] or NULL (= 1);
 * element arithmetic in float64 (lgamma / digamma / log / pow), as the reference's promoted expression.                     */
DH_API int dh_zinb_nll_forward_f32(int64_t n, int64_t n_genes, const float* X, int64_t ldx, const float* mean, int64_t ldm,
                            const float* disp, int64_t ldd, const float* pi, int64_t ldp, const double* scale_factor,
                            double ridge_lambda, double* rowloss, dh_stream_t stream);
DH_API int dh_zinb_nll_backward_f32(int64_t n, int64_t n_genes, const float* X, int64_t ldx, const float* mean, int64_t ldm,
                             const float* disp, int64_t ldd, const float* pi, int64_t ldp, const double* scale_factor,
                             double ridge_lambda, const double* upstream, float* d_mean, float* d_disp, float* d_pi, int64_t ldo,
                             dh_stream_t stream);
/* The same loss as a function of the decoder heads' RAW outputs: the heads are `Sequential(Linear, MeanAct())`, `Sequential(Linear,
 * DispAct())`, `Sequential(Linear, Sigmoid())` (scdsc.py:409-411 with MeanAct / DispAct at :601-618; sctag.py:531-548), i.e.
 * mean = clamp(exp(a), 1e-5, 1e6), disp = clamp(softplus(a), 1e-4, 1e4), pi = sigmoid(a) — 5 forward and ~12 backward elementwise
 * passes over N x G matrices when left to torch.  Here the activations and their Jacobians (torch's backward formulas: inclusive clamp
 * bounds, softplus threshold 20) are evaluated inside the two loss kernels; d_*_raw are the gradients w.r.t. the raw outputs.        */
DH_API int dh_zinb_nll_logits_forward_f32(int64_t n, int64_t n_genes, const float* X, int64_t ldx, const float* mean_raw, int64_t ldm,
                                   const float* disp_raw, int64_t ldd, const float* pi_raw, int64_t ldp, const double* scale_factor,
                                   double ridge_lambda, double* rowloss, dh_stream_t stream);
DH_API int dh_zinb_nll_logits_backward_f32(int64_t n, int64_t n_genes, const float* X, int64_t ldx, const float* mean_raw, int64_t ldm,
                                    const float* disp_raw, int64_t ldd, const float* pi_raw, int64_t ldp, const double* scale_factor,
                                    double ridge_lambda, const double* upstream, float* d_mean_raw, float* d_disp_raw, float* d_pi_raw,
                                    int64_t ldo, dh_stream_t stream);

/* scDSC's joint loop (scdsc.py:265-283) evaluates this loss and its gradient on the same operands back to back, and the three heads'
 * bias gradients are the gradients' column sums: ONE pass instead of three (forward 32 GB + backward 56 GB + column sums 24 GB at
 * 1M x 2000).  On return mean_raw / disp_raw / pi_raw (leading dimension ld) hold unit * d(sum of element losses) / d(raw output) —
 * `unit` is the caller's constant (1 / (N G) for the mean), the run-time upstream scalar is folded by the caller into the heads' dW / db
 * — loss_partials[n_loss] the float64 partial sums of the element losses (sum them in index order) and col_partials
 * [n_row_blocks][3][n_genes] the per-row-block column sums of the three gradients (dh_colsum_f32 over the [n_row_blocks, 3 n_genes]
 * matrix gives d bias).  dh_zinb_heads_fused_partials returns the two counts (host arithmetic).                                   */
DH_API int dh_zinb_heads_fused_partials(int64_t n, int64_t n_genes, int64_t* n_loss, int64_t* n_row_blocks);
DH_API int dh_zinb_heads_fused_f32(int64_t n, int64_t n_genes, const float* X, int64_t ldx, float* mean_raw, float* disp_raw, float* pi_raw,
                            int64_t ld, const double* scale_factor, double ridge_lambda, double unit, double* loss_partials,
                            float* col_partials, dh_stream_t stream);

/* ---- multi-GPU: RCCL over xGMI, one process per GPU ------------------------------------------------
 * The reference has no multi-GPU path for these models; the sharded layer replaces the single-process torch.spmm / autograd
 * pair of scdsc.py:498 + :286-288 (spagcn.py:359 + :521) with a destination-range shard per rank and ONE exchange per SpMM.
 * RCCL is bound at run time (dlopen) on the first call: no link-time dependency, and inside a PyTorch process the RCCL torch
 * already loaded is the one used.  Bootstrap: rank 0 calls dh_comm_unique_id and ships the 128 bytes to the other ranks by any
 * host channel (a file, MPI, torch.distributed's store); every rank then calls dh_comm_init.  Collectives are asynchronous on
 * the given stream.  Row counts of the halo exchange are HOST arrays of `world` entries (this rank's entry 0).
 *   dh_comm_allgather_rows_f32: out[r * rows_per_rank ...] = rank r's `local` [rows_per_rank, width] (dense exchange of S / G).
 *   dh_comm_allreduce_f32:      buf <- sum over ranks (dW, db: 4 MB at the headline shape).
 *   dh_comm_halo_exchange_f32:  all-to-all-v of contiguous row blocks, `width` floats per row, ordered by peer rank: grouped
 *                               ncclSend / ncclRecv, every pair in flight at once on the point-to-point xGMI links.
 *   dh_comm_halo_spmm_f32:      Y[rows] = act(A_local * operand + bias) for a shard whose column ids are [own rows | halo rows]:
 *                               packs operand[send_idx] (times the ReLU sign mask of those rows if send_relu_mask != NULL) into
 *                               send_buf, exchanges on comm_stream while the interior rows run on compute_stream, then the
 *                               boundary rows; `operand` is [n_local + n_halo, ldz], its tail receives the halo.  Rows travel with
 *                               the operand's own stride: send_buf holds sum(send_rows) * ldz floats (ldz >= width; the padding
 *                               columns of a padded operand ride along so that the halo lands row for row).  No host sync. */
#define DH_COMM_UNIQUE_ID_BYTES 128
typedef struct dh_comm* dh_comm_t;
DH_API int dh_comm_unique_id(void* id_host /* DH_COMM_UNIQUE_ID_BYTES */);
DH_API int dh_comm_init(dh_comm_t* comm, int world, int rank, const void* unique_id_host);
DH_API int dh_comm_destroy(dh_comm_t comm);
DH_API int dh_comm_world(dh_comm_t comm);
DH_API int dh_comm_rank(dh_comm_t comm);
DH_API int dh_comm_allgather_rows_f32(dh_comm_t comm, const float* local, int64_t rows_per_rank, int64_t width, float* out,
                               dh_stream_t stream);
DH_API int dh_comm_allreduce_f32(dh_comm_t comm, float* buf, int64_t count, dh_stream_t stream);
/* Host arithmetic of one all-to-all-v (no GPU, no communicator): offsets (in rows) of peer p's block in the packed send buffer and in
 * the receive buffer, blocks ordered by peer rank, totals in *n_send / *n_recv (may be NULL); rejects negative counts and a non-zero
 * count for `rank` itself.  dh_comm_halo_exchange_f32 issues exactly one ncclSend(send + send_offset[p] * width, send_rows[p] * width)
 * and one ncclRecv(recv + recv_offset[p] * width, recv_rows[p] * width) per peer with a non-zero count, inside one group. */
DH_API int dh_comm_halo_offsets(int world, int rank, const int64_t* send_rows_host, const int64_t* recv_rows_host,
                         int64_t* send_offset_host, int64_t* recv_offset_host, int64_t* n_send, int64_t* n_recv);
DH_API int dh_comm_halo_exchange_f32(dh_comm_t comm, const float* send, const int64_t* send_rows_host, float* recv,
                               const int64_t* recv_rows_host, int64_t width, dh_stream_t stream);
DH_API int dh_comm_halo_spmm_f32(dh_comm_t comm, int64_t n_local, int64_t n_halo, int64_t width,
                               const int32_t* rowptr, const int32_t* col, const float* val, float* operand, int64_t ldz,
                               const int32_t* send_idx, const int64_t* send_rows_host, const int64_t* recv_rows_host, float* send_buf,
                               const int32_t* interior_rows, int64_t n_interior, const int32_t* boundary_rows, int64_t n_boundary,
                               float* Y, int64_t ldy, const float* bias, int act, const void* send_relu_mask,
                               dh_stream_t compute_stream, dh_stream_t comm_stream);

/* ---- optimiser step of the captured mini-batch loops --------------------------------------------------------------------------------
 * torch.optim.Adam (amsgrad = False) over n fp32 tensors given as HOST arrays of device pointers (params, grads, exp_avg, exp_avg_sq, the
 * per-tensor step counters — one float each, incremented here — and element counts): two launches per 8 tensors instead of the ~14
 * multi-tensor launches (or one 38 us fused launch) of the framework optimiser; arithmetic in torch's single-tensor order, every
 * operation rounded separately in fp32.  dance: scdeepsort.py:160, graphsc.py:180 (Adam for every mini-batch model).                      */
DH_API int dh_adam_step_f32(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     float* const* step, const int64_t* numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                     dh_stream_t stream);

/* ---- the persistent mini-batch steps (ministep.hip) ---------------------------------------------------------------------------------
 * One call = a run of consecutive training steps of the reference's mini-batch loops at its default batch sizes, 4 launches per step,
 * no host read, nothing but kernels on the stream.  The kernels walk the seed cells' rows of the CellFeatureGraph-layout CSR directly
 * (genes are nodes [0, n_genes), a cell row = its genes ascending + its self loop): no block, no renumbering, no transposed copy.
 * Adam state is torch.optim.Adam's own (exp_avg, exp_avg_sq, one float32 step counter per tensor, all on the device), updated in place.
 * Dropout masks are Philox4x32-10 draws keyed by (seed, step0 + step index, layer, element).
 * bad[0] collects flags (never cleared here): 1 = a seed is not a cell row of that layout, 2 = a seed without exactly one self loop
 * (graph-sc's identity decoder target), 4 = a label outside [0, n_classes).
 * phase: 0 = whole steps; 1 = stop after the gradients, written to `grads` (flat: w1 | b1 | w2 | b2) — the data-parallel form, the
 * caller all-reduces them; 2 = apply `grads` (Adam only); 3 (graph-sc) = the aggregation alone: ax_out[2, batch, in_feats] = the
 * D_out^-1/2 / edge-weight / D_in^-1/2 sums of both forwards (each with its own dropout draw) — what the layer multiplies by its weight;
 * large batches run the dense layers and the all-pairs decoder on the big-tile kernels from there.  Phases 1 - 3 take n_steps == 1.      */
typedef struct dh_adam_state { /* one parameter tensor of torch.optim.Adam (amsgrad = False) */
  float* param;
  float* exp_avg;
  float* exp_avg_sq;
  float* step;
} dh_adam_state_t;

/* graph-sc: dance/modules/single_modality/clustering/graphsc.py:181-230 (the batch loop of GraphSC.fit), :274-383 (GCNAE: one
 * WeightedGraphConv in_feats -> hidden norm="both" + ReLU, one Linear hidden -> emb), :386-411 (inner-product decoder), :414-484.
 * Per step s (seeds[s * batch .. + batch)): z_out[s * batch + i] = the embedding of the FIRST forward (:202-203), loss_out[s] = the
 * loss of the second (:215-216), then Adam (:217-219).  Aggregate-first order (AX, then (AX) W1): see ministep.hip.                      */
typedef struct dh_graphsc_step {
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  const float* features; /* [n_nodes, in_feats] */
  int64_t ld_features, n_nodes, n_genes;
  int64_t batch, in_feats, hidden, emb;
  int32_t agg_mean; /* fn.mean instead of fn.sum (:463-465) */
  int32_t phase;
  int32_t max_row_entries; /* the largest in-degree of a cell row of the graph (0 = unknown): short rows let phase 3 run on the matrix cores */
  int32_t reserved;
  dh_adam_state_t w1, b1, w2, b2; /* layer1.weight [in_feats, hidden], layer1.bias, encoder.0.weight [emb, hidden], encoder.0.bias */
  float lr, beta1, beta2, eps, weight_decay;
  float dropout, decoder_dropout;
  uint64_t seed, step0;
  const int64_t* seeds;
  float* z_out;
  float* loss_out;
  int32_t* bad;
  float* grads;
  float* ax_out;
  void* workspace;
  size_t workspace_bytes;
} dh_graphsc_step_t;
DH_API int dh_graphsc_step_supported(int64_t batch, int64_t in_feats, int64_t hidden, int64_t emb);
DH_API size_t dh_graphsc_step_workspace_bytes(int64_t n_genes, int64_t batch, int64_t in_feats, int64_t hidden, int64_t emb);
DH_API int dh_graphsc_steps(const dh_graphsc_step_t* cfg_host, int64_t first_step, int64_t n_steps, dh_stream_t stream);

/* scDeepSort: dance/modules/single_modality/cell_type_annotation/scdeepsort.py:222-257 (cal_loss), :26-88 (GNN: one AdaptiveSAGE
 * dim_in -> hidden + ReLU, Linear hidden -> n_classes), dance/models/nn/gnn.py:62-96 (the layer output is Linear(dropout(h_dst)); the
 * weighted-mean aggregation is computed and dropped: neigh_out != NULL keeps computing it — [batch, dim_in] of the last step —,
 * NULL skips it), :185 (CrossEntropyLoss(reduction="sum")).  loss_out[s] = the summed loss of step s.                                    */
typedef struct dh_scdeepsort_step {
  const int32_t* rowptr;
  const int32_t* col;
  const float* val;
  const void* features; /* [n_nodes, dim_in] f32, or bf16 when features_bf16 */
  int64_t ld_features, n_nodes, n_genes;
  const int32_t* cell_id; /* [n_nodes]: gene index, -1 for cells */
  const int64_t* labels;  /* [n_nodes] */
  const float* alpha;     /* [n_genes + 2] */
  int64_t batch, dim_in, hidden, n_classes;
  int32_t features_bf16;
  int32_t phase;
  dh_adam_state_t w1, b1, w2, b2; /* layers.0.layers.1.weight [hidden, dim_in], .bias, linear.weight [n_classes, hidden], .bias */
  float lr, beta1, beta2, eps, weight_decay;
  float dropout;
  uint64_t seed, step0;
  const int64_t* seeds;
  float* loss_out;
  float* neigh_out;
  int32_t* bad;
  float* grads;
  void* workspace;
  size_t workspace_bytes;
} dh_scdeepsort_step_t;
DH_API int dh_scdeepsort_step_supported(int64_t batch, int64_t dim_in, int64_t hidden, int64_t n_classes);
DH_API size_t dh_scdeepsort_step_workspace_bytes(int64_t batch, int64_t dim_in, int64_t hidden, int64_t n_classes);
DH_API int dh_scdeepsort_steps(const dh_scdeepsort_step_t* cfg_host, int64_t first_step, int64_t n_steps, dh_stream_t stream);

/* The dropout draw of the two loops as a standalone launch (tests, known-answer checks of the generator): out[e] = 0 or 1 / (1 - p)
 * for element e of stream `sid` at step `step`.                                                                                          */
DH_API int dh_ministep_dropout_mask_f32(int64_t n, float p, uint64_t seed, uint64_t step, int32_t sid, float* out, dh_stream_t stream);

/* ---- the DEC heads' target distribution and KL loss (dec_loss.hip) ---------------------------------------------------------------------
 * dance/modules/spatial/spatial_domain/spagcn.py:398-407 (loss_function: mean over spots of sum_j p log(p / (q + 1e-6))), :421-425
 * (target_distribution: p = q^2 / sum_i q, rows normalised); GC_DEC :609-620; scDSC's target_distribution scdsc.py:431-441.
 * colsum_q [c] = the column sums of q (dh_colsum_f32; all-reduced by the caller when the spots are sharded).  kl_forward: *loss =
 * scale * sum_ij p log(p / (q + eps)) (scale = 1 / n_spots for the mean), double accumulation in a fixed order; kl_backward:
 * dq = -(g ? *g : 1) * scale * p / (q + eps), g = the upstream gradient of the scalar loss, on the device.                                */
DH_API int dh_dec_target_f32(int64_t n, int64_t c, const float* q, int64_t ldq, const float* colsum_q, float* p, int64_t ldp, dh_stream_t stream);
DH_API size_t dh_dec_kl_workspace_bytes(void);
DH_API int dh_dec_kl_forward_f32(int64_t n, int64_t c, const float* p, int64_t ldp, const float* q, int64_t ldq, float eps, double scale, float* loss,
                          void* workspace, size_t workspace_bytes, dh_stream_t stream);
DH_API int dh_dec_kl_backward_f32(int64_t n, int64_t c, const float* p, int64_t ldp, const float* q, int64_t ldq, float eps, double scale, const float* g,
                           float* dq, int64_t ldd, dh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DANCE_HIP_H */
