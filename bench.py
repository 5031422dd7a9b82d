#!/usr/bin/env python
"""Headline benchmark: cells/sec through one GCN layer forward+backward (scDSC GNNLayer semantics,
2000 genes -> 512, fp32) on a synthetic 1M-cell x 2k-gene k=15 graph (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = relu(A (X W)) forward, then backward from a fixed random dY producing dW (X is a non-grad leaf, as
in the reference: scdsc.py:247,286-288).  With N > 1 the 1M cells are sharded by destination range
(dance_amd/sharding.py): strong scaling; the transformed features / output gradients are exchanged over RCCL
(all-gather, or the feature-sliced all-to-all when the graph has no locality) and dW is all-reduced, all inside
the timed region.  Inputs are generated on the device and resident in HBM before
the timed region; graph set-up (CSR transpose) is outside it, as graph construction is in the reference.
Prints ONE JSON line on rank 0.  Besides the headline (rand-k15) the line carries

* ``roofline``: the dominant kernel against its own roofline AND ``layer_hbm_frac`` = the layer's algorithmic bytes
  (85.9 GB at the headline size) / step time / 8 TB/s — the metric's "fraction of HBM roofline";
* ``gemm_f32x3_row`` (1 GPU): a separately labelled row, NOT the headline — the same layer with the GEMMs on the bf16
  matrix cores via an exact three-way operand split (dh_gemm_f32x3);
* ``knn_k15`` (1 GPU): the same layer timed on the metric's literal graph — exact kNN (k = 15, self included) of a
  clustered 50-d embedding + UMAP connectivities, built by NeighborGraph's own kernels inside this script, with the cells
  renumbered by a locality order at graph set-up (``unordered``: the same graph as built; outputs identical row for row);
* N > 1: ``exchange`` — every exchange mode timed with the same K steps (halo all-to-all-v, dense all-gather, feature-sliced
  all-to-all), bytes on the wire per step and the exchange alone timed without compute; the headline is the fastest mode.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CELLS, N_GENES, N_HIDDEN, K_NEIGH = 1_000_000, 2000, 512, 15
PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PEAK_MFMA_F32_TFLOPS = 157.3  # dense f32-input MFMA peak


def synth_features(n_rows, n_genes, device, seed, kind="expression"):
    """The layer's input X, made on the device (set-up, untimed).
    ``expression`` = SURVEY.md §8(d)'s generator: cells from 20 Gaussian clusters in a 50-d latent space pushed through a fixed random
    50 x G map that modulates per-gene rates lambda_g ~ LogNormal(0, 1); counts ~ Poisson(rate) masked by Bernoulli(0.10); then
    normalize_total(1e4) -> log1p -> per-gene standardisation (zero mean, unit variance over this rank's rows) clipped at 10, as
    scdsc.py:124-126 prepares it.  The cluster centres, the map and lambda_g come from a fixed seed (shared by all ranks); the cells
    from ``seed``.  ``randn`` = standard normal entries: the A/B that shows the matrix-core rate is not an artefact of the ~90 %
    constant entries of a standardised sparse matrix (MFMA power is data dependent)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n_rows, n_genes), dtype=torch.float32, device=device)
    step = 125_000
    if kind == "randn":
        for lo in range(0, n_rows, step):
            hi = min(n_rows, lo + step)
            out[lo:hi] = torch.randn((hi - lo, n_genes), device=device, generator=g)
        return out
    gs = torch.Generator(device=device).manual_seed(20240)  # structure shared by every rank
    centres = torch.randn((20, 50), device=device, generator=gs) * 2.0
    proj = torch.randn((50, n_genes), device=device, generator=gs) / 50**0.5
    lam = torch.exp(torch.randn(n_genes, device=device, generator=gs))
    s1 = torch.zeros(n_genes, dtype=torch.float64, device=device)
    s2 = torch.zeros(n_genes, dtype=torch.float64, device=device)
    for lo in range(0, n_rows, step):
        hi = min(n_rows, lo + step)
        z = centres[torch.randint(0, 20, (hi - lo, ), device=device, generator=g)] + torch.randn((hi - lo, 50), device=device, generator=g)
        rate = lam[None, :] * torch.exp(0.3 * (z @ proj))
        counts = torch.poisson(rate, generator=g) * (torch.rand((hi - lo, n_genes), device=device, generator=g) < 0.10)
        tot = counts.sum(1, keepdim=True).clamp_(min=1.0)
        x = torch.log1p(counts / tot * 1e4)
        s1 += x.sum(0, dtype=torch.float64)
        s2 += (x.double() * x.double()).sum(0)
        out[lo:hi] = x
    mean = s1 / n_rows
    std = (s2 / n_rows - mean * mean).clamp_(min=1e-12).sqrt()
    mean, inv = mean.float(), (1.0 / std).float()
    for lo in range(0, n_rows, step):
        hi = min(n_rows, lo + step)
        out[lo:hi] = ((out[lo:hi] - mean) * inv).clamp_(max=10.0)
    return out


def _mix64(x):
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around; logical shifts emulated with masks)."""
    lsr = lambda z, sh: (z >> sh) & ((1 << (64 - sh)) - 1)
    z = x + (-7046029254386353131)                # 0x9E3779B97F4A7C15
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)  # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def synth_rand_graph_rows(n, k, lo, hi, device, seed):
    """Rows [lo, hi) of 'rand-k15' (SURVEY.md §8d): k DISTINCT uniformly random in-neighbours != i per row i, sorted, value 1/k.
    Counter-based: row i's columns are a pure function of (seed, i) — a rank generates exactly its own destination range and gets the
    rows the whole graph would have there, so no rank of a P-GPU run ever builds (or stores) the other ranks' rows.  k + 3 hashed
    candidates in [0, n - 1), the first k distinct ones (in generation order) taken, sorted and shifted past the diagonal; a row with fewer than k distinct
    candidates (1e-10 at 1M x 15) is re-hashed with the next salt."""
    rows = torch.arange(lo, hi, device=device, dtype=torch.int64)
    m = k + 3
    out = torch.empty((hi - lo, k), dtype=torch.int64, device=device)
    todo = torch.arange(hi - lo, device=device)
    salt = 0
    while todo.numel():
        ctr = (rows[todo, None] * m + torch.arange(m, device=device)[None, :]) + ((seed * 1024 + salt) << 40)
        c = ((_mix64(ctr) >> 1) & ((1 << 62) - 1)) % (n - 1)
        srt = c.sort(dim=1, stable=True)
        dup_sorted = torch.zeros_like(c, dtype=torch.bool)
        dup_sorted[:, 1:] = srt.values[:, 1:] == srt.values[:, :-1]
        dup = torch.zeros_like(dup_sorted).scatter_(1, srt.indices, dup_sorted)  # back in generation order: the later copy is the duplicate
        # the first k distinct candidates IN GENERATION ORDER (taking the k smallest would bias the columns towards 0), then sorted
        keep = torch.argsort(dup.to(torch.int8), dim=1, stable=True)[:, :k]
        ok = ~torch.gather(dup, 1, keep).any(dim=1)
        c = torch.gather(c, 1, keep).sort(dim=1).values
        out[todo[ok]] = c[ok]
        todo = todo[~ok]
        salt += 1
    out = out + (out >= rows[:, None]).to(torch.int64)  # skip the diagonal (order is preserved)
    rowptr = torch.arange(0, (hi - lo) * k + 1, k, device=device, dtype=torch.int32)
    val = torch.full(((hi - lo) * k,), 1.0 / k, dtype=torch.float32, device=device)
    return rowptr, out.to(torch.int32).reshape(-1).contiguous(), val


def synth_rand_graph(n, k, device, seed):
    """The whole 'rand-k15' graph (one GPU, and the "alltoall" exchange mode, which replicates the CSR by design)."""
    return synth_rand_graph_rows(n, k, 0, n, device, seed)


def synth_knn_graph(n, k, device, seed, reorder=True, order="morton"):
    """'knn-k15' (SURVEY.md §8d): cells from 20 Gaussian clusters in a 50-d latent space; exact kNN (self included) and
    UMAP connectivities by the NeighborGraph kernels (dh_knn_bruteforce_f32, dh_umap_membership_f32, ...).  With ``reorder`` the
    cells are then renumbered by a locality order (``order``: Z-order over the embedding's principal components on the device, or
    reverse Cuthill-McKee on the host; dance_amd.graph.locality_order — graph set-up, like the kNN search itself) and the graph permuted with the edge order inside every row kept, so the layer's outputs are those of
    the unordered graph row for row, bit for bit.  Returns (graph as built, renumbered graph or None, perm, build s, order s)."""
    from dance_amd import kernels
    from dance_amd.graph import CSRGraph, locality_order
    g = torch.Generator(device=device).manual_seed(seed)
    centers = torch.randn((20, 50), device=device, generator=g) * 4.0
    emb = centers[torch.randint(0, 20, (n, ), device=device, generator=g)] + torch.randn((n, 50), device=device, generator=g)
    def build():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, dist_ = kernels.knn(emb, k)
        out = kernels.umap_connectivities(idx, dist_.contiguous())[0]
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0
    # built twice, the second (steady) time reported: a process's first build also pays the first launches of a dozen kernels and the
    # first allocation of a multi-GB workspace — 0.18 - 0.41 s over the boxes of round 6 for the same 0.16 s of kernels
    _, build_cold_s = build()
    (rowptr, col, val), build_s = build()
    synth_knn_graph.cold_build_s = build_cold_s
    graph = CSRGraph(rowptr, col, val, n, n, symmetric=True)
    if not reorder:
        return graph, None, None, build_s, (0.0, 0.0)
    t0 = time.perf_counter()
    # "morton": Z-order over the leading principal components of the embedding the neighbours were searched in, on the device
    # (dance_amd.graph.morton_order); "rcm": reverse Cuthill-McKee of the pattern on the host (0.5 s at 1M cells)
    perm = (locality_order(graph, "morton", coords=emb) if order == "morton" else locality_order(graph)).to(device)
    torch.cuda.synchronize()
    order_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ordered = graph.permute(perm)
    torch.cuda.synchronize()
    permute_s = time.perf_counter() - t0
    # (rounds 3-4 reported order + permutation as ONE figure, round 5 the order alone under the same key — both are
    # reported now: locality_order_s = computing the order, locality_permute_s = renumbering the graph)
    return graph, ordered, perm, build_s, (order_s, permute_s)


def layer_bytes(n, nnz, f=N_GENES, h=N_HIDDEN, s=4):
    """B_layer of SURVEY.md §8(d): GEMM fwd + SpMM(A) + SpMM(A^T) + GEMM dW."""
    return 2.0 * n * f * s + 4.0 * n * h * s + 2.0 * nnz * h * s + 2.0 * nnz * (4 + s) + 8.0 * (n + 1) + 2.0 * f * h * s


def source_hashes():
    """sha256 of the kernel sources the PMC traffic file was collected for (stale counters must not be reported)."""
    import hashlib
    out = {}
    for name in ("gemm_f32.hip", "spmm.hip", "common.h"):
        with open(os.path.join(ROOT, "dance_amd", "csrc", name), "rb") as fh:
            out[name] = hashlib.sha256(fh.read()).hexdigest()
    return out


def cpu_baseline(sample_cells, min_seconds=10.0, max_iters=5):
    """The reference's CPU path for the same layer — oracle.layers.GNNLayer (torch-CPU mm + spmm + autograd,
    a port of scdsc.py:475-501; the AST-lifted reference class itself cannot travel to the GPU box) — timed on this host's
    cores on a bounded sample of the workload drawn by the GPU leg's own generators (SURVEY 8(d) expression X with the same
    seeds, rand-k15 rows, xavier W, fixed dY), down-sampled to ``sample_cells`` cells."""
    import numpy as np
    from oracle import layers as ol
    n = sample_cells
    cpu = torch.device("cpu")
    x = synth_features(n, N_GENES, cpu, seed=100)
    layer = ol.GNNLayer(N_GENES, N_HIDDEN)
    bound = (6.0 / (N_GENES + N_HIDDEN))**0.5
    with torch.no_grad():
        layer.weight.copy_((torch.rand((N_GENES, N_HIDDEN), generator=torch.Generator().manual_seed(2)) * 2 - 1) * bound)
    _, col, val = synth_rand_graph(n, K_NEIGH, cpu, seed=1)
    row = torch.arange(n).repeat_interleave(K_NEIGH)
    adj = torch.sparse_coo_tensor(torch.stack([row, col.long()]), val, (n, n))
    dy = torch.randn((n, N_HIDDEN), generator=torch.Generator().manual_seed(3))

    def step():
        layer.weight.grad = None
        layer(x, adj).backward(dy)

    step()  # warm-up (includes torch's COO coalesce bookkeeping on first use)
    times = []
    t_all = time.perf_counter()
    while len(times) < max_iters and (time.perf_counter() - t_all < min_seconds or len(times) < 2):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    full = None  # the same port on the WHOLE 1M-cell workload: minutes of CPU time, measured once per refresh and kept under profiles/
    fpath = os.path.join(ROOT, "profiles", "cpu_baseline_1M.json")
    if os.path.exists(fpath):
        fj = json.load(open(fpath))
        full = {"value": fj["value"], "cores": fj["cores"], "source": "profiles/cpu_baseline_1M.json (bench.py --cpu-sample-cells 1000000 on a GPU box)"}
    return {"value": n / med, "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port", "full_workload_value": full,
            "sample": f"{n} cells x {N_GENES} genes -> {N_HIDDEN} drawn by the GPU leg's generators (8(d) expression X, rand-k{K_NEIGH} graph, "
                      f"same seeds), fp32, fwd+bwd, median of {len(times)} iterations ({med * 1e3:.0f} ms each); kind 'port' = "
                      f"oracle.layers.GNNLayer on torch-CPU (a restatement of scdsc.py:475-501, not the reference class itself: "
                      f"/root/reference does not exist on the bench box)",
            "host_cpu_count": os.cpu_count()}


def time_steps(step, fence, steps, warmup, world, dev, timer_cls):
    """W untimed + exactly K timed steps bracketed by barrier + synchronize; MAX over ranks.  Returns (seconds, timer)."""
    for _ in range(warmup):
        step()
    fence()
    with timer_cls() as timer:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, timer


def time_exchange_only(sg, h, iters, fence, dev):
    """The collectives of one step (forward + backward exchange) on buffers of the real size, nothing else: ms per step."""
    n_local = sg.a.n_rows
    s = torch.randn((n_local, h), device=dev)

    def once():
        if sg.mode == "halo":
            for plan in (sg.halo, sg.halo_t):
                recv = torch.empty((plan.n_halo, h), device=dev)
                w = sg.halo_exchange(plan, s[plan.send_idx.long()] if plan.send_idx.numel() else s[:0], recv)
                if w is not None:
                    w.wait()
        elif sg.mode == "alltoall":
            for _ in range(2):
                sg.columns_to_rows(sg.rows_to_columns(s), n_local)
        else:
            sg.all_gather_rows(s)
            sg.all_gather_rows(s)
    saved = dict(sg.stats)  # the dry exchanges below must not show up in the layer's own accounting
    once()
    fence()
    warm = sg.stats["exchanged_bytes"]
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    fence()
    ms = (time.perf_counter() - t0) / iters * 1e3
    t = torch.tensor([ms], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_step = (sg.stats["exchanged_bytes"] - warm) / iters
    sg.stats.update(saved)
    return float(t.item()), per_step


def run_config_rows():
    """The model-level rows, every group in its OWN process (scripts/bench_configs.py <row names>): the host-bound rows (graph-sc's
    large-batch loop is ~45 launches per 1.3 ms batch) measured 13 - 26 % slower when they ran late in one long process — after the other
    rows' CPU legs and 100 GB of allocator traffic — than in a fresh one (c4: 160 ms alone, 181 after c3, 202 at the end of the line;
    same kernel time in all three: profiles/r06final_*).  A fit is what a user runs in a process; that is what a row times."""
    import subprocess
    groups = [["c2_gcn_100k"], ["c2_scdsc_epoch_100k", "c2_scdsc_epoch_1M"], ["c3_scdeepsort_1M_bf16_epoch"], ["c4_graphsc_1M_epoch_1gpu"], ["c5_spagcn_500k_iter"]]
    rows = {}
    for names in groups:
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_configs.py"), *names], capture_output=True, text=True, timeout=900)
            rows.update(json.loads(r.stdout[r.stdout.index("{"):]))
        except Exception as e:  # noqa: BLE001 — reported in place; the headline line must survive
            for nm in names:
                rows[nm] = {"error": f"{type(e).__name__}: {e}"}
    rows["_note"] = "every group of rows ran in its own process (python scripts/bench_configs.py <rows>)"
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=N_CELLS, help="total cells (default: the BASELINE config)")
    ap.add_argument("--cpu-sample-cells", type=int, default=100_000,
                    help="cells of the CPU baseline sample (1000000 = the full workload: ~1.5 min of host time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn-workload", action="store_true", help="skip the second (knn-k15) timed workload at 1 GPU")
    ap.add_argument("--no-x3-row", action="store_true", help="skip the separately labelled split-bf16 GEMM row at 1 GPU")
    ap.add_argument("--no-x-randn", action="store_true", help="skip the A/B leg with standard-normal X at 1 GPU")
    ap.add_argument("--locality", choices=["morton", "rcm"], default="morton",
                    help="locality order of the knn-k15 workload: Z-order over the embedding's principal components on the device (default) or "
                         "reverse Cuthill-McKee on the host")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the model-level rows of BASELINE configs 2, 3 and 5 (scripts/bench_configs.py; ~1.5 min at 1 GPU)")
    ap.add_argument("--exchange", choices=["auto", "halo", "allgather", "alltoall"], default="auto",
                    help="multi-GPU exchange (dance_amd/sharding.py); auto = time every mode, headline = the fastest")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="p",
                    help="with --of P: time rank p's shard of a P-GPU run on this one GPU and PROJECT the step time "
                         "(scripts/emulate_rank.py; prints its own JSON line, labelled a projection)")
    ap.add_argument("--of", type=int, default=8, metavar="P")
    args = ap.parse_args()
    if args.emulate_rank is not None:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import emulate_rank
        return emulate_rank.main(["--cells", str(args.cells), "--ranks-of", str(args.of), "--rank", str(args.emulate_rank)])

    from dance_amd import _lib, kernels, sharding
    from dance_amd.graph import CSRGraph

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    _lib.require_device()
    # Development switches (never set by the driver): DANCE_AMD_BENCH_ONE_GPU=1 puts every rank on GPU 0 and DANCE_AMD_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device) — the only way to run the P > 1 code path, with the real kernels, streams and
    # events, on a one-GPU box.  Such a line is labelled "backend": "gloo ..." and is a functional check, not a measurement.
    backend = os.environ.get("DANCE_AMD_BENCH_BACKEND", "nccl")
    if os.environ.get("DANCE_AMD_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    n = args.cells

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- inputs resident in HBM before the timed region ---------------------------------------------------
    ranges, _ = sharding.row_ranges(n, world)
    lo, hi = ranges[rank]
    n_local = hi - lo
    # every rank generates ITS destination rows of the graph only (counter-based generator: the rows the whole graph has there); the
    # rows of A^T come from one set-up exchange (sharding.transpose_shard).  Only the "alltoall" mode, which replicates the CSR by
    # design, builds the whole graph.
    graph = None

    def whole_graph():
        nonlocal graph
        if graph is None:
            graph = CSRGraph(*synth_rand_graph(n, K_NEIGH, dev, seed=1), n, n)
        return graph

    def make_sharded(mode):
        if world == 1 or mode == "alltoall":
            return sharding.ShardedGCNGraph.from_global_csr(whole_graph(), mode=mode)
        rp, cl, vl = synth_rand_graph_rows(n, K_NEIGH, lo, hi, dev, seed=1)
        return sharding.ShardedGCNGraph.from_row_shard(sharding.GraphShard(rp, cl, vl, lo, hi, n), n, mode=mode)

    x = synth_features(n_local, N_GENES, dev, seed=100 + rank)
    gen = torch.Generator(device=dev).manual_seed(2)
    bound = (6.0 / (N_GENES + N_HIDDEN))**0.5  # xavier_uniform, same W on every rank
    w = ((torch.rand((N_GENES, N_HIDDEN), device=dev, generator=gen) * 2 - 1) * bound).requires_grad_(True)
    dy = torch.randn((n_local, N_HIDDEN), device=dev, generator=torch.Generator(device=dev).manual_seed(3 + rank))
    torch.cuda.synchronize()

    def make_step(sg):
        def step():
            w.grad = None
            y = sharding.sharded_gcn_layer(x, w, sg, None, True)  # (x is read at call time: the randn A/B swaps it)
            y.backward(dy)
        return step

    modes = [args.exchange]
    if world == 1:
        modes = ["allgather"]  # one GPU: the shard is the graph, no exchange
    elif args.exchange == "auto":
        modes = ["halo", "allgather"] + (["alltoall"] if N_HIDDEN % world == 0 else [])
    runs, failed = {}, {}
    for mode in modes:
        # one exchange mode that raises (on every rank alike: a plan or shape error) must not cost the whole line when several are
        # being compared; every rank learns of a failure anywhere before the mode's figures are used
        err = None
        try:
            sg = make_sharded(mode)  # world == 1: the whole graph
            elapsed, timer = time_steps(make_step(sg), fence, args.steps, args.warmup, world, dev, kernels.KernelTimer)
            runs[mode] = dict(sg=sg, elapsed=elapsed, ksum=timer.summary(), bytes_per_step=sg.stats["exchanged_bytes"] / max(args.steps + args.warmup, 1))
            if world > 1:
                ex_ms, ex_bytes = time_exchange_only(sg, N_HIDDEN, max(3, args.steps // 4), fence, dev)
                runs[mode].update(exchange_only_ms=ex_ms, exchange_only_bytes=ex_bytes)
        except Exception as e:  # noqa: BLE001 — reported in the JSON line; re-raised below when no mode is left
            if len(modes) == 1:
                raise
            err = f"{type(e).__name__}: {e}"
            sg = None
        if world > 1 and len(modes) > 1:
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                failed[mode] = err or "failed on another rank"
                runs.pop(mode, None)
        if len(modes) > 1:
            if mode in runs:  # (a failed mode has no entry)
                runs[mode]["sg"] = None
            del sg
            torch.cuda.empty_cache()
    if not runs:
        raise SystemExit(f"every exchange mode failed: {failed}")
    mode = min(runs, key=lambda m: runs[m]["elapsed"])
    elapsed, ksum = runs[mode]["elapsed"], runs[mode]["ksum"]
    sg = runs[mode]["sg"] or make_sharded(mode)
    nnz_total = K_NEIGH * n
    comm_info = None
    if world > 1:
        # what the communicator itself saw (not what the launcher was asked for): ranks, backend, every rank's device
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev),
                "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))}
        comm_info = {"ranks": dist.get_world_size(), "backend": dist.get_backend()}
        try:  # (a diagnostic: it must never cost the line)
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            comm_info.update(devices=gathered, distinct_devices=len({(g["device"], g["uuid"]) for g in gathered}))
        except Exception as e:  # noqa: BLE001
            comm_info["devices_error"] = f"{type(e).__name__}: {e}"

    x3_out = None
    if world == 1 and not args.no_x3_row:
        # NOT the headline: the same layer with both GEMMs on the bf16 matrix cores (dh_gemm_f32x3: every fp32 operand split
        # exactly into three bf16 terms, 6 partial products, fp32 accumulation; gated at the exact kernel's error vs float64 by
        # tests/test_gpu_gemm_x3.py).  The headline above computes in fp32 on the f32-input matrix cores (dtype "f32").
        prev = kernels.GEMM_MODE
        kernels.GEMM_MODE = "x3"
        try:
            x_elapsed, x_timer = time_steps(make_step(sg), fence, args.steps, args.warmup, world, dev, kernels.KernelTimer)
        finally:
            kernels.GEMM_MODE = prev
        x_ms = x_elapsed / args.steps * 1e3
        x3_out = {"label": "separate row, not the headline: GEMMs by dh_gemm_f32x3 (fp32 in / fp32 out, bf16 x 3 operand split, "
                           "6 products, fp32 accumulate; error vs float64 <= the exact kernel's, tests/test_gpu_gemm_x3.py)",
                  "ms_per_step": x_ms, "value": n / (x_elapsed / args.steps), "unit": "cells/s",
                  "kernels_ms": {k: round(v[1], 4) for k, v in sorted(x_timer.summary().items())}}

    randn_out = None
    if world == 1 and not args.no_x_randn:
        # A/B of the input DATA, same shape, same kernels: standard-normal X instead of the standardised expression matrix
        x_keep = x
        x = synth_features(n_local, N_GENES, dev, seed=900, kind="randn")
        try:
            r_elapsed, r_timer = time_steps(make_step(sg), fence, args.steps, args.warmup, world, dev, kernels.KernelTimer)
        finally:
            x = x_keep
        randn_out = {"label": "A/B, not the headline: the same layer with X ~ N(0, 1) entries (matrix-core power is data dependent)",
                     "ms_per_step": r_elapsed / args.steps * 1e3, "value": n / (r_elapsed / args.steps), "unit": "cells/s",
                     "kernels_ms": {k: round(v[1], 4) for k, v in sorted(r_timer.summary().items())}}
        del x_keep

    knn_out = None
    if world == 1 and not args.no_knn_workload:
        del sg
        kg, kg_ordered, perm, build_s, order_s = synth_knn_graph(n, K_NEIGH, dev, seed=7, order=args.locality)
        sgk = sharding.ShardedGCNGraph.from_global_csr(kg)
        k_elapsed, _ = time_steps(make_step(sgk), fence, args.steps, args.warmup, world, dev, kernels.KernelTimer)
        k_ms = k_elapsed / args.steps * 1e3
        y_plain = sharding.sharded_gcn_layer(x, w, sgk, None, True).detach()
        del sgk
        # the same cells renumbered by locality: X and dY are permuted ONCE (set-up, like the graph build), the layer is unchanged
        x_saved, dy_saved = x, dy
        x, dy = x_saved[perm].contiguous(), dy_saved[perm].contiguous()
        sgo = sharding.ShardedGCNGraph.from_global_csr(kg_ordered)
        o_elapsed, o_timer = time_steps(make_step(sgo), fence, args.steps, args.warmup, world, dev, kernels.KernelTimer)
        o_ms = o_elapsed / args.steps * 1e3
        y_ord = sharding.sharded_gcn_layer(x, w, sgo, None, True).detach()
        same = bool(torch.equal(y_ord, y_plain[perm]))
        x, dy = x_saved, dy_saved
        del y_ord, y_plain, sgo
        knn_out = {"workload": f"same layer on knn-k15: exact kNN (k={K_NEIGH}, self included) of a 20-cluster 50-d embedding + UMAP "
                               f"connectivities, built on the device by the NeighborGraph kernels; cells renumbered by "
                               f"{'Z-order over the 3 leading principal components of the embedding (device)' if args.locality == 'morton' else 'reverse Cuthill-McKee (host)'}"
                               f" (graph set-up), X permuted once, outputs identical row for row",
                   "locality_order": args.locality,
                   "nnz": int(kg.nnz), "graph_build_s": round(build_s, 4), "graph_build_first_call_s": round(getattr(synth_knn_graph, "cold_build_s", 0.0), 4), "locality_order_s": round(order_s[0], 4), "locality_permute_s": round(order_s[1], 4),
                   "locality_note": "locality_order_s = computing the order only (as BENCH_r05 reported it); rounds 3-4 quoted order + permutation as one figure",
                   "ms_per_step": o_ms, "value": n / (o_elapsed / args.steps), "unit": "cells/s",
                   "layer_hbm_frac": round(layer_bytes(n, kg.nnz) / (o_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                   "y_bit_identical_to_unordered": same,
                   "kernels_ms": {k_: round(v[1], 4) for k_, v in sorted(o_timer.summary().items())},
                   "unordered": {"ms_per_step": k_ms, "value": n / (k_elapsed / args.steps)}}
        del kg, kg_ordered
        sg = make_sharded(mode)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        nnz_local = int(sg.a.col.numel())
        nnz_t_local = int(sg.at.col.numel())
        gemm_flops = 2.0 * n_local * N_GENES * N_HIDDEN

        sliced = world > 1 and mode == "alltoall"  # every rank aggregates ALL rows over H / world columns
        width = N_HIDDEN // world if sliced else N_HIDDEN
        if sliced:
            nnz_local = nnz_t_local = int(sg.full[0].col.numel())

        def spmm_bytes(nnz, rows):  # B_gather of SURVEY.md §8d: indices+values, row pointers, gathered rows, output
            rows = n if sliced else rows
            return nnz * 8.0 + 4.0 * (rows + 1) + nnz * width * 4.0 + rows * width * 4.0

        kernels_out = {}
        for name, (launches, ms) in sorted(ksum.items()):
            e = {"launches": launches, "avg_ms": round(ms, 4)}
            if name.startswith("gemm_f32"):
                e.update(bound="mfma", achieved=round(gemm_flops / ms / 1e9, 2), peak=PEAK_MFMA_F32_TFLOPS, unit="TFLOP/s")
            elif name.startswith("spmm_csr_f32") and (world == 1 or mode != "halo"):  # halo mode: two partial launches per call
                b = spmm_bytes(nnz_local if "fwd" in name else nnz_t_local, n_local)
                e.update(bound="hbm", achieved=round(b / ms / 1e6, 1), peak=PEAK_HBM_GBS, unit="GB/s")
            elif name.startswith("relu_backward"):
                e.update(bound="hbm", achieved=round(3.0 * n_local * N_HIDDEN * 4 / ms / 1e6, 1), peak=PEAK_HBM_GBS, unit="GB/s")
            elif name.startswith("relu_mask_apply"):  # dY in, masked dY out, one bit per element of mask
                e.update(bound="hbm", achieved=round((2.0 * 4 + 0.125) * n_local * N_HIDDEN / ms / 1e6, 1), peak=PEAK_HBM_GBS, unit="GB/s")
            if "achieved" in e:
                e["frac"] = round(e["achieved"] / e["peak"], 4)
            kernels_out[name] = e
        dominant = max((k for k in kernels_out if "achieved" in kernels_out[k]), key=lambda k: kernels_out[k]["avg_ms"] * kernels_out[k]["launches"])
        d = kernels_out[dominant]
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # PMC-derived bytes/launch, see profiles/README.md
        if os.path.exists(tpath) and n == N_CELLS and world == 1:
            tj = json.load(open(tpath))
            if tj.get("source_sha256") == source_hashes():
                traffic = tj.get(dominant)
            else:
                traffic_note = "profiles/hbm_traffic.json was collected for other kernel sources (sha256 mismatch): not reported"
        roofline = {"kernel": dominant, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                    "unit": d["unit"], "frac": d["frac"], "traffic": traffic,
                    "layer_algorithmic_GB": round(layer_bytes(n, nnz_total) / 1e9, 2),
                    "layer_hbm_frac": round(layer_bytes(n, nnz_total) / (ms_per_step * 1e-3) / (PEAK_HBM_GBS * 1e9), 4)}
        if traffic_note:
            roofline["traffic_note"] = traffic_note
        out = {
            "metric": "cells/sec per GCN fwd+bwd, 1M cells x 2k genes k=15",
            "value": n / (elapsed / args.steps), "unit": "cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"GCN layer (scDSC GNNLayer) fwd+bwd, {n} cells x {N_GENES} genes -> {N_HIDDEN}, "
                                   f"rand-k{K_NEIGH} graph (nnz={K_NEIGH * n}), fp32; X = SURVEY 8(d) generator (20-cluster latent x "
                                   f"LogNormal gene rates, Poisson counts at 10 % density, normalize_total / log1p / per-gene scale)",
                       "cells": n, "genes": N_GENES, "hidden": N_HIDDEN, "k": K_NEIGH,
                       "parallelism": f"dst-range x{world}, {mode} exchange" if world > 1 else "single GPU",
                       **({"exchange_mode": mode, "exchange_selection": (f"--exchange {args.exchange}: " + ("every mode timed with the same K steps in this run, the fastest is the headline"
                                                                         if args.exchange == "auto" else "as requested")),
                           "rccl": comm_info} if world > 1 else {})},
            "roofline": roofline, "kernels": kernels_out,
        }
        if backend != "nccl":
            out["backend"] = f"{backend} (development run, all ranks on one GPU: NOT a measurement)"
        if world > 1:
            out["exchange"] = {m: {"ms_per_step": round(r["elapsed"] / args.steps * 1e3, 4), "value": n / (r["elapsed"] / args.steps),
                                   "bytes_on_wire_per_step_per_rank": int(r["exchange_only_bytes"]),
                                   "exchange_only_ms_per_step": round(r["exchange_only_ms"], 4)} for m, r in runs.items()}
            out["exchange"]["headline_mode"] = mode
            if failed:
                out["exchange"]["failed_modes"] = failed
        if x3_out is not None:
            out["gemm_f32x3_row"] = x3_out
        if randn_out is not None:
            out["x_randn"] = randn_out
        if knn_out is not None:
            out["knn_k15"] = knn_out
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_cells)
        if world == 1 and not args.no_configs and n == N_CELLS:
            # model-level rows of the other BASELINE configs (2: ScDSC epoch at 100k / 1M, 3: ScDeepSort bf16 epoch, 5: SpaGCN iteration at
            # 500k spots), each with its own roofline and CPU baseline; never the headline, never fatal
            del x, dy, sg, w
            torch.cuda.empty_cache()
            out["configs"] = run_config_rows()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
