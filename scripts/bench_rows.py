"""Per-row measurement of the SURVEY.md §8 kernels at BASELINE-config sizes: GPU time (HIP events), algorithmic
bytes / ops, roofline fraction, and the CPU oracle (or the library the reference calls) timed on a bounded sample.

    python scripts/bench_rows.py [--quick] > profiles/rNN_rows.json

Not the headline bench (that is bench.py); this feeds the kernel table of DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

HBM, VALU_OPS = 8000.0, 78.65  # GB/s peak; T non-fused f32 ops/s (157.3 TFLOP/s counts an FMA as 2)
dev = torch.device("cuda:0")


def gpu_ms(fn, iters=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def cpu_s(fn, iters=1):
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--skip-knn", action="store_true", help="skip the (seconds-long) kNN rows")
    ap.add_argument("--knn-only", action="store_true", help="only the kNN / UMAP rows")
    args = ap.parse_args()
    q = args.quick
    from oracle import graphs as og
    from oracle import matrix as om
    from oracle import sage as osg
    rows = {}
    g = torch.Generator(device=dev).manual_seed(0)

    # ---- K8 exact kNN (A11/A12/A13) ------------------------------------------------------------------------
    for n, d, k in ([] if args.skip_knn else [(20_000, 50, 15)] if q else [(100_000, 50, 15), (1_000_000, 50, 15), (100_000, 2000, 15)]):
        x = torch.randn(n, d, device=dev, generator=g)
        ops = 3.0 * n * n * d
        ns = 4000
        xs = x[:ns].cpu().numpy()
        t_or = cpu_s(lambda: og.knn_exact(xs, k))
        from sklearn.neighbors import NearestNeighbors
        t_sk = cpu_s(lambda: NearestNeighbors(n_neighbors=k, algorithm="brute").fit(xs).kneighbors(xs))
        cpu = dict(sample=f"{ns} points (pairs scale n^2)", oracle_numpy_pairs_per_s=ns * ns / t_or,
                   sklearn_brute_pairs_per_s=ns * ns / t_sk, cores=os.cpu_count())
        # the exact scan (vector ALUs, 3 ops per pair and feature) and the matrix-core filter + exact re-rank: same output
        ms = gpu_ms(lambda: kernels.knn(x, k, algo=kernels.KNN_SCAN), iters=1, warm=1 if n <= 100_000 else 0)
        rows[f"knn_bruteforce_f32 [scan] n={n} d={d} k={k}"] = dict(
            ms=ms, bound="valu", achieved=ops / ms / 1e9, peak=VALU_OPS, unit="Tops/s (sub,mul,add)", frac=ops / ms / 1e9 / VALU_OPS,
            cells_per_s=n / ms * 1e3, gpu_pairs_per_s=n * n / ms * 1e3, cpu_baseline=cpu)
        with kernels.KernelTimer() as tm:
            ms_f = gpu_ms(lambda: kernels.knn(x, k, algo=kernels.KNN_FILTER), iters=2, warm=1)
        dp = (d + 7) // 8 * 8
        # d <= 64: one fp16 term + 6 threshold columns; d > 64: three bf16 terms (hi.hi + hi.lo + lo.hi)
        k3 = (dp + 6 + 15) // 16 * 16 if d <= 64 else (3 * dp + 15) // 16 * 16
        flops = 2.0 * n * n * k3
        rows[f"knn_bruteforce_f32 [filter] n={n} d={d} k={k}"] = dict(
            ms=ms_f, speedup_vs_scan=ms / ms_f, bound="mfma", achieved=flops / ms_f / 1e9, peak=2500.0, unit="TFLOP/s (fp16 / bf16 filter flops over the whole call)",
            frac=flops / ms_f / 1e9 / 2500.0, cells_per_s=n / ms_f * 1e3, gpu_pairs_per_s=n * n / ms_f * 1e3, cpu_baseline=cpu,
            note="sample scan + split + filter + re-rank; identical output to [scan] (tests/test_gpu_graphs.py)")
        if n == 1_000_000 or q:
            idx, dist = kernels.knn(x, k)
            ms_u = gpu_ms(lambda: kernels.umap_connectivities(idx, dist), iters=2)
            t_u = cpu_s(lambda: og.fuzzy_simplicial_set(*og.knn_exact(xs, k), k))
            rows[f"umap_connectivities n={n} k={k}"] = dict(ms=ms_u, cells_per_s=n / ms_u * 1e3,
                                                            cpu_baseline=dict(sample=f"{ns} cells incl. kNN", cells_per_s=ns / t_u))
        del x

    # spatial coordinates (d = 3, BASELINE config 5's 500k spots; a 1M-point 2-d layout): the cell-grid form (DH_KNN_GRID, what KNN_AUTO picks for d <= 3)
    for n, d, k in ([] if args.skip_knn else [(20_000, 3, 15)] if q else [(500_000, 3, 15), (1_000_000, 2, 15)]):
        x = torch.rand(n, d, device=dev, generator=g) * (n ** (1.0 / d))
        ms_g = gpu_ms(lambda: kernels.knn(x, k, algo=kernels.KNN_GRID), iters=3, warm=1)
        ms_f = gpu_ms(lambda: kernels.knn(x, k, algo=kernels.KNN_FILTER), iters=1, warm=1)
        i_g, d_g = kernels.knn(x, k, algo=kernels.KNN_GRID)
        i_f, d_f = kernels.knn(x, k, algo=kernels.KNN_FILTER)
        rows[f"knn_bruteforce_f32 [grid] n={n} d={d} k={k}"] = dict(
            ms=ms_g, filter_ms=ms_f, speedup_vs_filter=ms_f / ms_g, cells_per_s=n / ms_g * 1e3, identical_to_filter=bool(torch.equal(i_g, i_f) and torch.equal(d_g, d_f)),
            bound="latency / L2 (about 100 pair evaluations per query instead of n)",
            note="bounding box, cell edge by bisection on the device, counting sort into cells, ring search with an exact stop rule: 7 launches, no host read")
        del x

    if args.knn_only:
        print(json.dumps(rows, indent=1))
        return

    # ---- A14 pairwise distance (SpaGCNGraph) ---------------------------------------------------------------
    n = 4096 if q else 16384
    xyz = torch.rand(n, 3, device=dev, generator=g) * 100
    ms = gpu_ms(lambda: kernels.pairwise_distance(xyz, 0))
    byt = n * n * 4.0
    ns = 1500
    t_or = cpu_s(lambda: om.pairwise_distance(xyz[:ns].cpu().numpy(), 0))
    rows[f"pairwise_distance_f32 n={n} d=3"] = dict(ms=ms, bound="hbm", achieved=byt / ms / 1e6, peak=HBM, unit="GB/s", frac=byt / ms / 1e6 / HBM,
                                                    cpu_baseline=dict(sample=f"{ns} spots", pairs_per_s=ns * ns / t_or, gpu_pairs_per_s=n * n / ms * 1e3))
    dmat = kernels.pairwise_distance(xyz, 0)
    ms = gpu_ms(lambda: kernels.gaussian_kernel(dmat, 1.5, want_out=False, want_rowsum=True))
    rows[f"gaussian_kernel_f32 rowsums (calculate_p) n={n}"] = dict(ms=ms, bound="hbm", achieved=byt / ms / 1e6, peak=HBM, unit="GB/s", frac=byt / ms / 1e6 / HBM)
    del dmat

    # ---- A1 CellFeatureGraph build + A3 AdaptiveSAGE aggregation (full graph) ------------------------------
    n_cells, n_genes, dfeat = (100_000 if q else 1_000_000), 2000, 400
    per = 200  # 10 % density: 200 expressed genes per cell
    col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
    rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=dev)
    val_x = torch.rand(n_cells * per, device=dev, generator=g) + 0.5
    nnz = n_cells * per

    def build():
        rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
        vx, vt = kernels.csr_row_normalize(rp_x, val_x), kernels.csr_row_normalize(rp_t, val_t)
        return kernels.cellgene_graph_assemble(rp_x, col, vx, rp_t, col_t, vt, perm_t, n_cells, n_genes)

    ms = gpu_ms(build, iters=2)
    ns = 2000
    xs = np.zeros((ns, n_genes), np.float32)
    cc = col[:ns * per].cpu().numpy().reshape(ns, per)
    xs[np.arange(ns)[:, None], cc] = val_x[:ns * per].cpu().numpy().reshape(ns, per)
    t_or = cpu_s(lambda: og.cell_feature_graph(xs))
    rows[f"cellgene graph build (transpose+normalize+assemble) cells={n_cells} nnz={nnz}"] = dict(
        ms=ms, edges_per_s=(2 * nnz + n_cells + n_genes) / ms * 1e3, cells_per_s=n_cells / ms * 1e3,
        cpu_baseline=dict(sample=f"{ns} cells (vectorised numpy oracle; the reference loops over nodes in Python)", cells_per_s=ns / t_or))
    rowptr, gcol, gval, eid = build()
    n_nodes = n_cells + n_genes
    feats = torch.randn(n_nodes, dfeat, device=dev, generator=g)
    cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
    alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5
    # cell <- gene aggregation (what scDeepSort's cell batches need): rows [G, G+N) of the CSR, gathered operand =
    # the 3.2 MB gene-feature table (cache resident), SURVEY.md §8d: B = nnz (4+4) + 4 (N+1) + G D 4 + N D 4
    rp_cells, cid_cells = rowptr[n_genes:], cid[n_genes:]
    ms = gpu_ms(lambda: kernels.sage_aggregate(rp_cells, gcol, gval, cid, cid_cells, alpha, feats), iters=3)
    e = nnz + n_cells
    byt = e * 8.0 + 4.0 * (n_cells + 1) + n_genes * dfeat * 4.0 + n_cells * dfeat * 4.0
    rows[f"sage_aggregate_f32 cell<-gene full graph cells={n_cells} D={dfeat} edges={e}"] = dict(
        ms=ms, bound="hbm", achieved=byt / ms / 1e6, peak=HBM, unit="GB/s", frac=byt / ms / 1e6 / HBM, cells_per_s=n_cells / ms * 1e3,
        l2_gather_GBs=e * dfeat * 4.0 / ms / 1e6)
    ms_all = gpu_ms(lambda: kernels.sage_aggregate(rowptr, gcol, gval, cid, cid, alpha, feats), iters=1)
    rows[f"sage_aggregate_f32 all nodes (gene rows gather {nnz} cell rows) cells={n_cells}"] = dict(
        ms=ms_all, note="dominated by the 2000 gene rows of ~1e5 in-edges each handled by one wavefront per row")
    # ---- config C3: the same aggregation and the dense update with bf16 storage ------------------------------
    feats16 = feats.to(torch.bfloat16)
    ms16 = gpu_ms(lambda: kernels.sage_aggregate_bf16(rp_cells, gcol, gval, cid, cid_cells, alpha, feats16), iters=3)
    byt16 = e * 8.0 + 4.0 * (n_cells + 1) + n_genes * dfeat * 2.0 + n_cells * dfeat * 2.0
    rows[f"sage_aggregate_bf16 cell<-gene full graph cells={n_cells} D={dfeat} edges={e}"] = dict(
        ms=ms16, bound="hbm", achieved=byt16 / ms16 / 1e6, peak=HBM, unit="GB/s", frac=byt16 / ms16 / 1e6 / HBM,
        cells_per_s=n_cells / ms16 * 1e3, l2_gather_GBs=e * dfeat * 2.0 / ms16 / 1e6)
    # the same aggregation on the matrix cores (dh_sage_window_mfma: adjacency densified per workgroup in LDS, bf16 hi + lo
    # splits; dense-equivalent flops 2 N G D per product plane, 3 planes for fp32 features, 2 for bf16)
    for tag, ft, planes in (("f32", feats, 3), ("bf16", feats16, 2)):
        msm = gpu_ms(lambda: kernels.sage_aggregate_mfma(rp_cells, gcol, gval, cid, cid_cells, alpha, ft, 0, n_genes), iters=3)
        fl = 2.0 * n_cells * n_genes * dfeat * planes
        rows[f"sage_window_mfma {tag} cell<-gene full graph cells={n_cells} D={dfeat} edges={e}"] = dict(
            ms=msm, bound="mfma", achieved=fl / msm / 1e9, peak=2500.0, unit="TFLOP/s (bf16 MFMA, dense-equivalent incl. the split planes)",
            frac=fl / msm / 1e9 / 2500.0, cells_per_s=n_cells / msm * 1e3,
            speedup_vs_gather=(ms if tag == "f32" else ms16) / msm)
    # the other direction: the 2000 gene rows of ~1e5 cell in-edges each (dh_sage_window_splitk: the cells as the K dimension, split
    # over the chip); the plan (one repack of the rows, graph only) is cached across calls and timed apart
    rp_genes, cid_genes = rowptr[:n_genes + 1], cid[:n_genes]
    for tag, ft, planes in (("f32", feats, 3), ("bf16", feats16, 2)):
        fng = lambda: kernels.sage_aggregate_splitk(rp_genes, gcol, gval, cid, cid_genes, alpha, ft, n_genes, n_cells)
        msm = gpu_ms(fng, iters=3)
        fl = 2.0 * n_cells * n_genes * dfeat * planes
        rows[f"sage_window_splitk {tag} gene<-cell full graph cells={n_cells} D={dfeat} edges={nnz + n_genes}"] = dict(
            ms=msm, bound="mfma", achieved=fl / msm / 1e9, peak=2500.0, unit="TFLOP/s (bf16 MFMA, dense-equivalent incl. the split planes)",
            frac=fl / msm / 1e9 / 2500.0, speedup_vs_all_nodes_gather=ms_all / msm,
            note="round 3: these rows cost 56.9 ms inside the all-nodes gather (one wavefront per row)")
    ms_all2 = gpu_ms(lambda: (kernels.sage_aggregate_splitk(rp_genes, gcol, gval, cid, cid_genes, alpha, feats, n_genes, n_cells),
                              kernels.sage_aggregate_mfma(rp_cells, gcol, gval, cid, cid_cells, alpha, feats, 0, n_genes)), iters=3)
    rows[f"AdaptiveSAGE aggregation of ALL nodes f32 (split-K gene rows + gene-window cell rows) cells={n_cells}"] = dict(
        ms=ms_all2, speedup_vs_all_nodes_gather=ms_all / ms_all2)
    PEAK_BF16 = 2500.0  # TFLOP/s dense (MI355X_MICROARCH.md)
    hid = 200
    h16 = feats16[n_genes:]  # [n_cells, 400] cell rows
    w16 = (torch.randn(hid, dfeat, device=dev, generator=g) / 20).to(torch.bfloat16)
    bias = torch.zeros(hid, device=dev)
    dy16 = torch.randn(n_cells, hid, device=dev, generator=g).to(torch.bfloat16)
    for name, fn, flops in [
        ("fwd  y = relu(h W^T + b)  [NT]", lambda: kernels.gemm_bf16(h16, w16, trans_b=True, bias=bias, act=kernels.ACT_RELU), 2.0 * n_cells * dfeat * hid),
        ("bwd  dh = g W             [NN]", lambda: kernels.gemm_bf16(dy16, w16), 2.0 * n_cells * dfeat * hid),
        ("bwd  dW = g^T h     [TN, split-K]", lambda: kernels.gemm_bf16(dy16, h16, trans_a=True, out_dtype=torch.float32), 2.0 * n_cells * dfeat * hid),
    ]:
        msg = gpu_ms(fn, iters=5, warm=2)
        rows[f"gemm_bf16 dense update {name} M={n_cells} in={dfeat} out={hid}"] = dict(
            ms=msg, bound="mfma", achieved=flops / msg / 1e9, peak=PEAK_BF16, unit="TFLOP/s", frac=flops / msg / 1e9 / PEAK_BF16,
            hbm_GBs=(n_cells * (dfeat + hid) * 2.0) / msg / 1e6)
    xb = torch.randn(n_cells, 2048, device=dev, generator=g).to(torch.bfloat16)
    wb = torch.randn(512, 2048, device=dev, generator=g).to(torch.bfloat16)
    msg = gpu_ms(lambda: kernels.gemm_bf16(xb, wb, trans_b=True), iters=5, warm=2)
    fl = 2.0 * n_cells * 2048 * 512
    rows[f"gemm_bf16 NT M={n_cells} K=2048 N=512 (kernel rate at a compute-heavy shape)"] = dict(
        ms=msg, bound="mfma", achieved=fl / msg / 1e9, peak=PEAK_BF16, unit="TFLOP/s", frac=fl / msg / 1e9 / PEAK_BF16)
    del xb, wb
    # ---- config C3 end to end: ScDeepSort.fit, one epoch (= train pass + the reference's two evaluation passes) on the
    # 1M-cell graph, large batches, fp32 vs bf16 storage -----------------------------------------------------------
    import tempfile
    from dance_amd.cellgraph import CellGeneGraph
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    feat_id = torch.cat((-torch.ones(n_genes, dtype=torch.int32), torch.arange(n_cells, dtype=torch.int32))).to(dev)
    cg = CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "feat_id": feat_id, "features": feats})
    labels = torch.randint(0, 16, (n_cells, ), generator=torch.Generator().manual_seed(0))
    bs = 65536
    for cd in ("fp32", "bf16"):
        with tempfile.TemporaryDirectory() as tmp:
            m = ScDeepSort(dfeat, 200, 1, "synthetic", "c3", batch_size=bs, device="cuda", save_root=tmp, verbose=False, compute_dtype=cd)
            torch.manual_seed(0)
            m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)  # warm-up epoch (allocator, lazy init)
            dt = None
            for _ in range(2):  # best of two: a fit call also writes a checkpoint to a temporary directory (one run in ten is 2x off)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with kernels.KernelTimer() as tm:
                    m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)
                    torch.cuda.synchronize()
                t = time.perf_counter() - t0
                if dt is None or t < dt:
                    dt = t
                    ks = {kname: [v[0], round(v[0] * v[1], 2)] for kname, v in tm.summary().items()}
        rows[f"ScDeepSort.fit 1 epoch (train + 2 eval passes) cells={n_cells} batch={bs} compute_dtype={cd}"] = dict(
            ms=dt * 1e3, cells_per_s=n_cells / dt, hip_kernels_calls_total_ms=ks,
            note="includes the block sampler (torch index ops) and checkpoint save per the reference's fit loop")
    # ---- config C4 on one GPU: GraphSC.fit, one epoch of the GAE (two forwards per batch as in graphsc.py:202,215), 50 -> 200 -> 300 ----
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    cg50 = cg.with_ndata(features=feats[:, :50].contiguous())
    for bsz in (8192, 128):  # 128 = the reference's batch size: the persistent step (dh_graphsc_steps; one C call per epoch)
        gs = GraphSC(in_feats=50, n_clusters=10, device="cuda")
        gs.fit(cg50, epochs=1, batch_size=bsz)  # warm-up

        def fit_s(epochs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gs.fit(cg50, epochs=epochs, batch_size=bsz)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        t1, t3 = fit_s(1), fit_s(3)
        dt = (t3 - t1) / 2  # an epoch inside a multi-epoch fit; fit(epochs=1) adds the capture and the read-out of z once
        rows[f"GraphSC.fit epoch cells={n_cells} batch={bsz} (reference default batch is 128)"] = dict(
            ms=dt * 1e3, cells_per_s=n_cells / dt, ms_per_batch=dt * 1e3 / -(-n_cells // bsz), fit_1_epoch_ms=t1 * 1e3, fit_3_epochs_ms=t3 * 1e3,
            once_per_fit_ms=(t1 - dt) * 1e3,
            step_mode=getattr(gs, "step_mode", None),
            note="epoch = (fit(epochs=3) - fit(epochs=1)) / 2.  step_mode 'ministep' (batches up to 512): every full batch of the epoch behind one C call, 4 "
            "launches per step; 'aggfirst' above: the aggregation off the CSR rows + dense layers / all-pairs decoder / Adam on the big-tile kernels.  once_per_fit = the "
            "capture (if any) + the read-out of the embedding to host numpy after the last epoch (1.2 GB at 1M cells, graphsc.py:232-236)")
    del feats, feats16, rowptr, gcol, gval, eid, cg, cg50

    # ---- SpaGCN-shape layer 50 -> 50 with bias, k = 15 -----------------------------------------------------
    from dance_amd.autograd import gcn_layer
    from dance_amd.graph import CSRGraph
    n = 100_000 if q else 1_000_000
    colk = torch.randint(0, n, (n, 15), device=dev, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
    graph = CSRGraph(torch.arange(0, n * 15 + 1, 15, dtype=torch.int32, device=dev), colk, torch.full((n * 15, ), 1 / 15., device=dev), n, n)
    graph.transpose()
    x50 = torch.randn(n, 50, device=dev, generator=g)
    w50 = (torch.randn(50, 50, device=dev, generator=g) / 7).requires_grad_(True)
    b50 = torch.zeros(50, device=dev, requires_grad=True)
    dy = torch.randn(n, 50, device=dev, generator=g)

    def step():
        w50.grad = b50.grad = None
        gcn_layer(x50, w50, graph, b50, False).backward(dy)

    ms = gpu_ms(step, iters=5, warm=2)
    byt = 1.85e3 * n
    byt_gather = n * 15 * (8 + 200.0) + 4 * n + 200.0 * n + 2 * 256.0 * n + 200.0 * n  # what the fused kernels actually have to move
    rows[f"GraphConvolution 50->50 fwd+bwd n={n} k=15"] = dict(ms=ms, bound="hbm", achieved=byt / ms / 1e6, peak=HBM, unit="GB/s",
                                                              frac=byt / ms / 1e6 / HBM, cells_per_s=n / ms * 1e3,
                                                              frac_gather_accounting=byt_gather / ms / 1e6 / HBM,
                                                              note="fused narrow layer (gcn_narrow.hip), X rows at stride 50")
    x64 = torch.zeros(n, 64, device=dev)
    x64[:, :50] = x50
    x50 = x64[:, :50]
    ms = gpu_ms(step, iters=5, warm=2)
    rows[f"GraphConvolution 50->50 fwd+bwd n={n} k=15, X rows 256-byte aligned (what SimpleGCDEC.fit feeds)"] = dict(
        ms=ms, bound="hbm", achieved=byt / ms / 1e6, peak=HBM, unit="GB/s", frac=byt / ms / 1e6 / HBM, cells_per_s=n / ms * 1e3,
        frac_gather_accounting=byt_gather / ms / 1e6 / HBM)
    del x64
    for (ns_, c_, d_, consts_, who) in ((n // 2, 10, 50, (0.2, 1e-8, 1.2, 0.5), "SpaGCN"), (n, 10, 32, (1.0, 0.0, 1.0, 1.0), "scDSC")):
        zz = torch.randn(ns_, d_, device=dev, generator=g)
        mu_ = torch.randn(c_, d_, device=dev, generator=g) * 0.7
        gq = torch.randn(ns_, c_, device=dev, generator=g)
        f_ms = gpu_ms(lambda: kernels.student_t_forward(zz, mu_, *consts_), iters=10, warm=2)
        b_ms = gpu_ms(lambda: kernels.student_t_backward(zz, mu_, *consts_, gq), iters=10, warm=2)
        sb = ns_ * (3.0 * d_ + 3.0 * c_) * 4
        rows[f"Student-t soft assignment fwd+bwd ({who} head) n={ns_} c={c_} d={d_}"] = dict(
            ms=f_ms + b_ms, forward_ms=f_ms, backward_ms=b_ms, bound="hbm", achieved=sb / (f_ms + b_ms) / 1e6, peak=HBM, unit="GB/s",
            frac=sb / (f_ms + b_ms) / 1e6 / HBM, note="bytes: forward N (d + c) 4, backward N (2 d + 2 c) 4")
        del zz, gq
    # ---- (f)2: fused ZINB NLL (fwd + bwd) and the all-pairs adjacency loss of scTAG without the N x N matrix -----------------------
    from dance_amd import autograd
    nz, gz = (50_000 if q else 1_000_000), 2000
    xr = torch.poisson(torch.rand(nz, gz, device=dev, generator=g) * 2)
    mean = (torch.rand(nz, gz, device=dev, generator=g) * 4 + 1e-3).requires_grad_(True)
    disp = (torch.rand(nz, gz, device=dev, generator=g) * 3 + 1e-3).requires_grad_(True)
    pi = (torch.rand(nz, gz, device=dev, generator=g) * 0.98 + 0.01).requires_grad_(True)
    sf = torch.rand(nz, device=dev, dtype=torch.float64) + 0.5

    def zstep():
        mean.grad = disp.grad = pi.grad = None
        autograd.zinb_nll(xr, mean, disp, pi, sf).backward()

    with kernels.KernelTimer() as tm:
        ms = gpu_ms(zstep, iters=3, warm=1)
    ks = {kname: round(v[1], 3) for kname, v in tm.summary().items()}
    zb = nz * gz * (16.0 + 28.0)
    rows[f"ZINB NLL fwd+bwd (dh_zinb_nll_*) cells={nz} genes={gz}"] = dict(ms=ms, kernels_ms=ks, bound="hbm", achieved=zb / ms / 1e6, peak=HBM, unit="GB/s",
                                                                          frac=zb / ms / 1e6 / HBM, note="float64 element arithmetic (3 lgamma, 2 digamma, 5 log, 1 pow): ALU-bound")
    # the same loss on the heads' raw outputs (MeanAct / DispAct / sigmoid and their backward inside the two kernels), against the
    # torch activations + their autograd around the loss kernels (what round 4 ran)
    am = torch.log(mean.detach()).requires_grad_(True)
    ad = torch.log(torch.expm1(disp.detach())).requires_grad_(True)
    ap = torch.logit(pi.detach()).requires_grad_(True)

    def zstep_logits():
        am.grad = ad.grad = ap.grad = None
        autograd.zinb_nll_from_logits(xr, am, ad, ap, sf).backward()

    def zstep_torch_acts():
        am.grad = ad.grad = ap.grad = None
        autograd.zinb_nll(xr, torch.clamp(torch.exp(am), 1e-5, 1e6), torch.clamp(torch.nn.functional.softplus(ad), 1e-4, 1e4), torch.sigmoid(ap), sf).backward()

    ms_l = gpu_ms(zstep_logits, iters=3, warm=1)
    ms_t = gpu_ms(zstep_torch_acts, iters=3, warm=1)
    rows[f"ZINB NLL on the heads' raw outputs fwd+bwd (dh_zinb_nll_logits_*) cells={nz} genes={gz}"] = dict(
        ms=ms_l, bound="hbm", achieved=zb / ms_l / 1e6, peak=HBM, unit="GB/s", frac=zb / ms_l / 1e6 / HBM, torch_activations_around_the_loss_kernels_ms=ms_t,
        note="same bytes as the row above (16 forward + 28 backward per element); the activations' 17 elementwise passes are gone")
    # round 6: loss + gradients (in place over the raw outputs) + the three bias column sums in one pass (dh_zinb_heads_fused_f32), against the
    # two logits kernels + three dh_colsum_f32 passes it replaces in ScDSC.fit
    raws = [t.detach().clone() for t in (am, ad, ap)]
    scratch = [r.clone() for r in raws]
    up = torch.tensor([1.0 / (nz * gz)], dtype=torch.float64, device=dev)

    def heads_three_pass():
        kernels.zinb_nll_forward(xr, *raws, sf, 0.0, logits=True)
        for dgrad in kernels.zinb_nll_backward(xr, *raws, sf, 0.0, up, logits=True):
            kernels.colsum(dgrad)

    def heads_fused():  # in place: restore the operands first (an 8 GB x 3 copy, outside the timed kernel: KernelTimer reads the kernel's own events)
        for dst, src in zip(scratch, raws):
            dst.copy_(src)
        kernels.zinb_heads_fused_(xr, *scratch, sf, 0.0, 1.0 / (nz * gz))

    ms_3 = gpu_ms(heads_three_pass, iters=3, warm=1)
    heads_fused()
    with kernels.KernelTimer() as tmf:
        for _ in range(3):
            heads_fused()
        torch.cuda.synchronize()
    ms_f = tmf.summary()["zinb_heads_fused_f32"][1]  # (launches, mean ms)
    hb = nz * gz * 28.0  # 4 x 4 bytes read + 3 x 4 written per element
    rows[f"ZINB heads: loss + gradients + bias sums in ONE pass (dh_zinb_heads_fused_f32) cells={nz} genes={gz}"] = dict(
        ms=ms_f, bound="hbm", achieved=hb / ms_f / 1e6, peak=HBM, unit="GB/s", frac=hb / ms_f / 1e6 / HBM, three_pass_ms=ms_3,
        note="28 bytes per element (x, three raw outputs read; three gradients written over them); three_pass_ms = dh_zinb_nll_logits_forward + _backward + 3 x dh_colsum_f32 on the same operands")
    del am, ad, ap, raws, scratch
    ns = 20_000 if q else 100_000  # the unfused torch formula (float64 temporaries) on a sample
    sl = slice(0, ns)
    import sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import cpu_ops as _ops

    def zref():
        m_, d_, p_ = (t[sl].detach().requires_grad_(True) for t in (mean, disp, pi))
        _ops._zinb_elements(xr[sl], m_, d_, p_, sf[sl], 0.0).mean().backward()

    ms_ref = gpu_ms(zref, iters=2, warm=1)
    rows[f"ZINB NLL fwd+bwd (dh_zinb_nll_*) cells={nz} genes={gz}"]["torch_unfused_float64_ms_scaled_to_full"] = ms_ref * nz / ns
    del xr, mean, disp, pi
    from dance_amd.graph import CSRGraph as _G
    for na in ([20_000] if q else [100_000, 1_000_000]):
        colk = torch.randint(0, na, (na, 15), device=dev, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
        ga = _G(torch.arange(0, na * 15 + 1, 15, dtype=torch.int32, device=dev), colk, torch.ones(na * 15, device=dev), na, na)
        z0 = (torch.randn(na, 32, device=dev, generator=g) * 0.2).requires_grad_(True)

        def astep():
            z0.grad = None
            autograd.adj_reconstruction_mse(z0, ga).backward()

        ms = gpu_ms(astep, iters=2, warm=1)
        fl = 4.0 * na * na * 64
        rows[f"scTAG adjacency loss over all N^2 pairs, no N x N matrix (adj_dim=32) N={na}"] = dict(
            ms=ms, bound="mfma", achieved=fl / ms / 1e9, peak=157.3, unit="TFLOP/s (fp32 matrix cores, d padded to 64)", frac=fl / ms / 1e9 / 157.3,
            note="dense N x N formulation: %.1f GB for the logits alone" % (na * na * 4 / 1e9))
        del ga, z0, colk
    # ---- A2 preprocessing on the device (opt-in): exact PCA of the gene x cell matrix + the cell-feature projection -------
    from dance_amd.utils.pca import pca_scores
    npc, gpc = (100_000 if q else 1_000_000), 2000
    xt = torch.rand(gpc, npc, device=dev, generator=g)          # genes x cells (WeightedFeaturePCA decomposes X^T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gene_feat, _, _ = pca_scores(xt, 400)
    torch.cuda.synchronize()
    t_pca = (time.perf_counter() - t0) * 1e3
    xrow = xt.t().contiguous()
    del xt
    t0 = time.perf_counter()
    cell_feat = kernels.gemm(xrow / xrow.sum(1, keepdim=True), gene_feat.contiguous())
    torch.cuda.synchronize()
    t_proj = (time.perf_counter() - t0) * 1e3
    rows[f"device PCA (WeightedFeaturePCA, opt-in) genes={gpc} cells={npc} k=400"] = dict(
        ms=t_pca + t_proj, pca_ms=t_pca, projection_ms=t_proj,
        note="Gram matrix over the cells on dh_gemm_f32 + float64 eigh of 2000 x 2000 (torch / rocSOLVER) + U S; then row-normalised X @ gene_feat")
    del xrow, cell_feat, gene_feat

    # ---- SURVEY.md §8d "knn-k15" (realistic locality): clustered cells -> exact kNN -> UMAP connectivities -> GCN layer ----
    del x50, w50, b50, dy, graph
    n, dlat, fin, hid = (100_000 if q else 1_000_000), 50, 2000, 512
    cent = torch.randn(20, dlat, device=dev, generator=g) * 4
    lat = cent[torch.randint(0, 20, (n, ), device=dev, generator=g)] + torch.randn(n, dlat, device=dev, generator=g)
    t0 = time.perf_counter()
    idx, dist = kernels.knn(lat, 15)
    (rp, cc, vv), _ = kernels.umap_connectivities(idx, dist)
    torch.cuda.synchronize()
    t_build = (time.perf_counter() - t0) * 1e3
    gk = CSRGraph(rp, cc, vv, n, n, symmetric=True)
    xk = torch.randn(n, fin, device=dev, generator=g)
    wk = (torch.randn(fin, hid, device=dev, generator=g) / 45).requires_grad_(True)
    dyk = torch.randn(n, hid, device=dev, generator=g)

    def step_k():
        wk.grad = None
        gcn_layer(xk, wk, gk, None, True).backward(dyk)

    with kernels.KernelTimer() as tm:
        ms = gpu_ms(step_k, iters=5, warm=2)
    ks = {kname: round(v[1], 3) for kname, v in tm.summary().items()}
    rows[f"GCN layer 2000->512 fwd+bwd on a REAL kNN graph (knn-k15, UMAP connectivities) n={n}"] = dict(
        ms=ms, cells_per_s=n / ms * 1e3, nnz=int(cc.numel()), graph_build_ms=t_build, kernels_ms=ks,
        note="graph built on the device by dh_knn_bruteforce_f32 (matrix-core filter) + the UMAP kernels; value-symmetric, so "
             "backward reuses the forward CSR; the headline bench uses the worst-case rand-k15 graph instead")
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
