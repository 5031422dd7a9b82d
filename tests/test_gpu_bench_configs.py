"""GPU: the model-level rows of bench.py's ``configs`` block (scripts/bench_configs.py) at toy sizes — the functions the driver's
bench run calls must keep working (a failure there would only show up as an ``error`` entry in the bench line)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def _check(row):
    assert "error" not in row, row
    assert row["ms"] > 0 and row["value"] > 0 and row["kernels_ms"] and row["cpu_baseline"]["value"] > 0 and row["cpu_baseline"]["kind"] == "port"
    assert row["roofline"]["kernel"] in row["kernels_ms"]


def test_config_rows_at_toy_sizes(cuda_device, monkeypatch):
    import bench
    import bench_configs as bc
    _check(bc.c5_spagcn_iter(cuda_device, n=20_000, e1=1, e2=4, cpu_spots=2_000))
    _check(bc.c2_scdsc_epoch(cuda_device, 4_000, 1, 3, cpu_sample=500))
    row = bc.c3_scdeepsort_epoch(cuda_device, n_cells=20_000, batch=4096, cpu_cells=2_000)
    _check(row)
    assert row["fp32"]["ms"] > 0
    _check(bc.c4_graphsc_epoch(cuda_device, n_cells=20_000, batch=4096, cpu_cells=2_000, ref_batch_epochs=False))
    monkeypatch.setattr(bc, "c2_gcn_100k", bc.c2_gcn_100k)  # (the 100k layer row runs at its real size: 4 ms a step)
    _check(bc.c2_gcn_100k(cuda_device, steps=3))
    assert bench.N_CELLS == 1_000_000  # the configs block only rides on the full-size headline run
