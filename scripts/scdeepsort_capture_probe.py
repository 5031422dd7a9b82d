"""GPU probe: does a captured training step pay at scDeepSort's LARGE batch (65536 cells, 1M-cell graph, bf16)?  The eager epoch has ~690
launches and a host read-back per block; steady-state epoch = (fit(4) - fit(1)) / 3, eager against captured
(DANCE_AMD_HIPGRAPH_MAX_BATCH=65536 DANCE_AMD_HIPGRAPH_MIN_BATCHES=1 in the environment of the captured run)."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import _cellgene_graph  # noqa: E402

from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n_cells = 1_000_000
    cg = _cellgene_graph(n_cells, 2000, 200, 400, dev)
    labels = torch.randint(0, 16, (n_cells, ), generator=torch.Generator().manual_seed(0))
    with tempfile.TemporaryDirectory() as tmp:
        m = ScDeepSort(400, 200, 1, "synthetic", "c3", batch_size=65536, device="cuda", save_root=tmp, verbose=False, compute_dtype="bf16")
        ts = {}
        for e in (1, 1, 4):
            torch.manual_seed(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.fit(cg, labels, epochs=e, lr=1e-3, val_ratio=0.2)
            torch.cuda.synchronize()
            ts[e] = time.perf_counter() - t0
        print(f"captured={m._use_graph} fit(1) {ts[1] * 1e3:.1f} ms  fit(4) {ts[4] * 1e3:.1f} ms  steady epoch {(ts[4] - ts[1]) / 3 * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
