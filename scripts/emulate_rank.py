#!/usr/bin/env python
"""One rank of a P-GPU run of the headline layer, measured on ONE GPU + a PROJECTION of the step time (no scaling curve is
measured here: the boxes available to the build have one GPU).

For P in (2, 4, 8) and rank p the real shard of the real graph is planned in this process (dance_amd.sharding, halo plan without
collectives), and every kernel that rank would run is timed at its real size: the local GEMM (N/P x 2000 x 512), the pack of the
rows the peers asked for, the interior and boundary SpMM of the forward, the mask pass, pack, interior / boundary SpMM of the
backward, the dW GEMM (split-K over N/P cells).  What cannot be measured — the exchange — is PROJECTED from the bytes each mode
puts on the wire and the xGMI figures of MI355X_MICROARCH.md (fully connected, one link per GPU pair, 153 GB/s per direction, all
P - 1 links of a GPU usable at once):
    halo       per exchange: max over peers of (rows from that peer x H x 4) / 153 GB/s
    allgather  per exchange: (N / P x H x 4) / 153 GB/s  (every peer sends its shard over its own link)
    all-reduce of dW (4 MB): a latency-bound ~60 us, counted once
    alltoall   per exchange (4 per step): (N / P x H / P x 4) / 153 GB/s per link; the SpMM then runs over ALL rows at width H / P
projected step = GEMM + pack + max(exchange, interior SpMM) + boundary SpMM   (forward)
               + mask + pack + max(exchange, interior SpMM) + boundary SpMM + dW GEMM + all-reduce   (backward)
alltoall step  = GEMM + mask + dW GEMM + all-reduce + 2 x (pack + unpack) + 4 exchanges + SpMM(A) + SpMM(A^T) over all rows, H / P wide
Graphs: rand-k15 (no locality) and knn-k15 renumbered by locality (Z-order over the embedding's principal components).
    python scripts/emulate_rank.py [--cells N] [--ranks-of 2,4,8] [--rank -1]   (-1: the middle rank)"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dance_amd import kernels, sharding  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402

LINK_GBS = 153.0  # one xGMI link, one direction (MI355X_MICROARCH.md)
ALLREDUCE_S = 60e-6


def t_ms(fn, it=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


def measure_rank(graph, rank, world, dev, F, H):
    sg = sharding.ShardedGCNGraph.from_global_csr(graph, mode="halo", emulate=(rank, world))
    lo, hi = sg.ranges[rank]
    n_loc = hi - lo
    x = bench.synth_features(n_loc, F, dev, seed=100 + rank)
    w = torch.randn((F, H), device=dev) * 0.02
    dy = torch.randn((n_loc, H), device=dev)
    r = {"rank": rank, "world": world, "rows": n_loc}
    out = {}
    for tag, plan, shard in (("fwd", sg.halo, sg.a), ("bwd", sg.halo_t, sg.at)):
        buf = torch.randn((n_loc + plan.n_halo, H), device=dev)
        y = torch.empty((n_loc, H), device=dev)
        mask = torch.empty(kernels.relu_mask_bytes(n_loc, H), dtype=torch.uint8, device=dev)
        e = {"halo_rows": plan.n_halo, "halo_frac_of_remote_rows": round(plan.n_halo / max(graph.n_rows - n_loc, 1), 4),
             "send_rows": int(plan.send_idx.numel()), "interior_rows": int(plan.interior.numel()), "boundary_rows": int(plan.boundary.numel()),
             "max_rows_from_one_peer": int(max(plan.recv_counts)) if plan.recv_counts else 0}
        if tag == "fwd":
            e["gemm_ms"] = t_ms(lambda: kernels.gemm(x, w, out=buf[:n_loc]))
            run = lambda rows: kernels.spmm_csr_relu(shard.rowptr, plan.col, shard.val, buf, n_cols=buf.shape[0], act=kernels.ACT_RELU,
                                                     out_mask=mask, out=y, rows=rows)
        else:
            kernels.spmm_csr_relu(sg.a.rowptr, sg.halo.col, sg.a.val, torch.randn((n_loc + sg.halo.n_halo, H), device=dev), n_cols=n_loc + sg.halo.n_halo,
                                  act=kernels.ACT_RELU, out_mask=mask, out=y)
            e["mask_ms"] = t_ms(lambda: kernels.relu_mask_apply(dy, mask, out=buf[:n_loc]))
            run = lambda rows: kernels.spmm_csr(shard.rowptr, plan.col, shard.val, buf, n_cols=buf.shape[0], out=y, rows=rows)
        e["pack_ms"] = t_ms(lambda: kernels.gather_rows(buf[:n_loc], plan.send_idx)) if plan.send_idx.numel() else 0.0
        e["interior_spmm_ms"] = t_ms(lambda: run(plan.interior)) if plan.interior.numel() else 0.0
        e["boundary_spmm_ms"] = t_ms(lambda: run(plan.boundary)) if plan.boundary.numel() else 0.0
        if tag == "bwd":
            e["gemm_dw_ms"] = t_ms(lambda: kernels.gemm(x, y, trans_a=True))
        e["halo_bytes_in"] = plan.n_halo * H * 4
        e["exchange_halo_ms_projected"] = e["max_rows_from_one_peer"] * H * 4 / (LINK_GBS * 1e9) * 1e3
        e["exchange_allgather_ms_projected"] = sg.chunk * H * 4 / (LINK_GBS * 1e9) * 1e3
        e["allgather_bytes_in"] = (world - 1) * sg.chunk * H * 4
        out[tag] = e
        del buf, y, mask
    f, b = out["fwd"], out["bwd"]
    compute = (f["gemm_ms"] + f["pack_ms"] + f["interior_spmm_ms"] + f["boundary_spmm_ms"] + b["mask_ms"] + b["pack_ms"] + b["interior_spmm_ms"] +
               b["boundary_spmm_ms"] + b["gemm_dw_ms"])
    halo = (f["gemm_ms"] + f["pack_ms"] + max(f["exchange_halo_ms_projected"], f["interior_spmm_ms"]) + f["boundary_spmm_ms"] + b["mask_ms"] +
            b["pack_ms"] + max(b["exchange_halo_ms_projected"], b["interior_spmm_ms"]) + b["boundary_spmm_ms"] + b["gemm_dw_ms"] + ALLREDUCE_S * 1e3)
    # dense all-gather: nothing to overlap with (every row may be needed): exchange, then one SpMM over all rows (= interior + boundary time)
    ag = compute - f["pack_ms"] - b["pack_ms"] + f["exchange_allgather_ms_projected"] + b["exchange_allgather_ms_projected"] + ALLREDUCE_S * 1e3
    # feature-sliced all-to-all (sharding.py mode "alltoall"): after the local GEMM every rank receives ALL rows of H / P columns,
    # aggregates all N rows at that width, and sends the result back; backward the same way round.  Measured here: the two pack /
    # unpack passes per exchange exactly as sharding.rows_to_columns / columns_to_rows perform them (minus the collective) and the
    # full-graph SpMM at width H / P on the real graph; projected: the all-to-all, (P - 1) x chunk x H / P x 4 bytes in over P - 1
    # links at once = chunk x H / P x 4 bytes per link.
    a2a = None
    if H % world == 0:
        hq, n_all = H // world, graph.n_rows
        gt = graph.transpose()
        loc = torch.randn((n_loc, H), device=dev)
        cols_buf = torch.randn((world * sg.chunk, hq), device=dev)
        ycols = torch.empty((n_all, hq), device=dev)

        def pack_rows_to_columns():
            send = torch.zeros((world, sg.chunk, hq), dtype=loc.dtype, device=dev)
            send[:, :n_loc] = loc.reshape(n_loc, world, hq).transpose(0, 1)
            return send

        def unpack_columns_to_rows():
            recv = cols_buf.reshape(world, sg.chunk, hq)
            return recv[:, :n_loc].transpose(0, 1).reshape(n_loc, world * hq).contiguous()
        a2a = {"slice_width": hq,
               "pack_ms": t_ms(pack_rows_to_columns), "unpack_ms": t_ms(unpack_columns_to_rows),
               "spmm_all_rows_fwd_ms": t_ms(lambda: kernels.spmm_csr(graph.rowptr, graph.col, graph.val, cols_buf[:n_all], n_cols=n_all, act=kernels.ACT_RELU, out=ycols)),
               "spmm_all_rows_bwd_ms": t_ms(lambda: kernels.spmm_csr(gt.rowptr, gt.col, gt.val, cols_buf[:n_all], n_cols=n_all, out=ycols)),
               "exchange_ms_projected_each": sg.chunk * hq * 4 / (LINK_GBS * 1e9) * 1e3,
               "bytes_in_per_exchange": (world - 1) * sg.chunk * hq * 4}
        a2a_step = (f["gemm_ms"] + b["mask_ms"] + b["gemm_dw_ms"] + ALLREDUCE_S * 1e3 + a2a["spmm_all_rows_fwd_ms"] + a2a["spmm_all_rows_bwd_ms"]
                    + 2 * (a2a["pack_ms"] + a2a["unpack_ms"]) + 4 * a2a["exchange_ms_projected_each"])
        del loc, cols_buf, ycols
    steps = {"halo": halo, "allgather": ag}
    if a2a is not None:
        steps["alltoall"] = a2a_step
    best = min(steps, key=steps.get)
    r.update(fwd=f, bwd=b, alltoall=a2a, measured_compute_ms=compute, projected_step_ms=steps, projected_best_mode=best,
             projected_cells_per_s={m: graph.n_rows / v * 1e3 for m, v in steps.items()})
    return r


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=bench.N_CELLS)
    ap.add_argument("--ranks-of", default="2,4,8")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--graphs", default="rand,knn-rcm")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    n, F, H, K = args.cells, bench.N_GENES, bench.N_HIDDEN, bench.K_NEIGH
    res = {"label": "PROJECTION, not a measured scaling curve: per-rank kernels measured on one MI355X at their P-GPU sizes, exchange "
                    "time from bytes on the wire at 153 GB/s per xGMI link (one link per GPU pair)", "cells": n, "link_GBs": LINK_GBS}
    graphs = {}
    if "rand" in args.graphs:
        rp, c, v = bench.synth_rand_graph(n, K, dev, seed=1)
        graphs["rand-k15"] = CSRGraph(rp, c, v, n, n)
    if "knn" in args.graphs:
        _, ordered, _, _, _ = bench.synth_knn_graph(n, K, dev, seed=7)
        graphs["knn-k15 (Z-order over the embedding's principal components)"] = ordered
    for gname, g in graphs.items():
        g.transpose()
        res[gname] = {"nnz": g.nnz}
        for world in [int(t) for t in args.ranks_of.split(",")]:
            rank = world // 2 if args.rank < 0 else min(args.rank, world - 1)
            r = measure_rank(g, rank, world, dev, F, H)
            res[gname][f"P={world}"] = r
            print(gname, f"P={world}", json.dumps({k: r[k] for k in ("measured_compute_ms", "projected_step_ms", "projected_best_mode")}), file=sys.stderr, flush=True)
            torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
