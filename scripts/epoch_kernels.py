"""Which kernels — the library's AND torch's — make up one epoch of a model's fit (ScDSC.fit: BASELINE config 2's model at 100k cells;
ScDeepSort.fit: config 3 at 1M cells, bf16)?  Run twice under `rocprofv3 --kernel-trace --stats` with different epoch counts and
difference the per-kernel totals:

    python scripts/epoch_kernels.py run scdsc 1          # one fit of 1 epoch (after a warm-up fit)
    python scripts/epoch_kernels.py run scdsc 7
    python scripts/epoch_kernels.py diff a_kernel_stats.csv b_kernel_stats.csv 6
"""
import csv
import os
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def run(epochs, n=100_000):
    import torch
    from bench_configs import _scdsc_inputs

    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    dev = torch.device("cuda", 0)
    x, counts, n_counts, graph, y = _scdsc_inputs(n, dev)
    graph.transpose()
    xh, ch, nh = x.cpu().numpy(), counts.cpu().numpy(), n_counts.cpu().numpy().astype("float64")
    del x, counts
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        m = ScDSC(pretrain_path=os.path.join(tmp, "ae.pt"), sigma=0.5, n_clusters=10, n_input=xh.shape[1], device="cuda")
        m.fit((graph, xh, ch, nh), y, lr=1e-3, epochs=1, pt_epochs=0)
        m.fit((graph, xh, ch, nh), y, lr=1e-3, epochs=epochs, pt_epochs=0)
        torch.cuda.synchronize()


def run_scdeepsort(epochs, n_cells=1_000_000, batch=65536):
    import torch
    from bench_configs import _cellgene_graph

    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    dev = torch.device("cuda", 0)
    cg = _cellgene_graph(n_cells, 2000, 200, 400, dev)
    labels = torch.randint(0, 16, (n_cells, ), generator=torch.Generator().manual_seed(0))
    with tempfile.TemporaryDirectory() as tmp:
        m = ScDeepSort(400, 200, 1, "synthetic", "c3", batch_size=batch, device="cuda", save_root=tmp, verbose=False, compute_dtype="bf16")
        torch.manual_seed(0)
        m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)
        m.fit(cg, labels, epochs=epochs, lr=1e-3, val_ratio=0.2)
        torch.cuda.synchronize()


def run_graphsc(epochs, n_cells=100_000, batch=128):
    import torch
    from bench_configs import _cellgene_graph

    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    dev = torch.device("cuda", 0)
    cg = _cellgene_graph(n_cells, 2000, 200, 50, dev)
    torch.manual_seed(0)
    gs = GraphSC(in_feats=50, n_clusters=10, device="cuda")
    gs.fit(cg, epochs=1, batch_size=batch)
    gs.fit(cg, epochs=epochs, batch_size=batch)
    torch.cuda.synchronize()


def diff(a, b, units):
    def load(p):
        return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
    ka, kb = load(a), load(b)
    rows = []
    for name, (cb, tb) in kb.items():
        ca, ta = ka.get(name, (0, 0.0))
        if cb != ca:
            rows.append(((tb - ta) / units / 1e6, (cb - ca) / units, name))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"per epoch: {total:.3f} ms of kernel time in {sum(r[1] for r in rows):.0f} launches")
    print("| ms / epoch | launches / epoch | kernel |\n|---|---|---|")
    for ms, calls, name in rows[:60]:
        print(f"| {ms:.4f} | {calls:.1f} | `{name[:150]}` |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        {"scdsc": run, "scdeepsort": run_scdeepsort, "graphsc": run_graphsc}[sys.argv[2]](int(sys.argv[3]), *(int(a) for a in sys.argv[4:6]))  # optional: cells (graphsc: cells, batch)
    else:
        diff(sys.argv[2], sys.argv[3], float(sys.argv[4]))
