"""The reference's integration strategy (tests/test_bench.py:171-204: ``runpy.run_path`` of every ``examples/**/<model>.py`` with light
options, asserting that the script does not raise), applied to this repo's drop-in examples — each one runs the model's own
``preprocessing_pipeline()`` on a device slot and then fit / predict / score, i.e. the seams between the transforms and the models."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LIGHT = {  # script -> (light options, lowest acceptable score: the synthetic programmes are well separated)
    "single_modality/cell_type_annotation/scdeepsort.py": (["--cells", "3000", "--genes", "400", "--dense_dim", "64", "--n_epochs", "12", "--batch_size", "256", "--lr", "0.01"], 0.85),
    "single_modality/cell_type_annotation/scheteronet.py": (["--cells", "1500", "--genes", "500", "--epochs", "25", "--use_zinb", "--cl_weight", "0.1"], 0.8),
    "single_modality/clustering/graphsc.py": (["--cells", "3000", "--genes", "600", "--nb_genes", "400", "--epochs", "2", "--batch_size", "128", "--in_feats", "30"], -1.0),
    "single_modality/clustering/scdsc.py": (["--cells", "1200", "--genes", "600", "--nb_genes", "300", "--topk", "15", "--epochs", "10", "--pretrain_epochs", "10"], 0.3),
    "single_modality/clustering/sctag.py": (["--cells", "1000", "--genes", "600", "--n_top_genes", "300", "--epochs", "10", "--pretrain_epochs", "15"], 0.3),
    # imputation: 1 - RMSE(imputed) / RMSE(zeros) on the held-out entries of the test cells (about +0.1 on these sparse synthetic counts)
    "single_modality/imputation/graphsci.py": (["--cells", "600", "--genes", "300", "--types", "3", "--n_epochs", "200", "--lr", "0.01"], -0.1),
    "spatial/spatial_domain/spagcn.py": (["--side", "24", "--genes", "300", "--epochs", "20", "--max_run", "4"], 0.3),
    "spatial/spatial_domain/stagate.py": (["--side", "24", "--genes", "400", "--high_variable_genes", "200", "--epochs", "40"], 0.3),
}


@pytest.mark.parametrize("script", sorted(LIGHT))
def test_example_runs(cuda_device, script):
    argv, floor = LIGHT[script]
    mod = runpy.run_path(os.path.join(ROOT, "examples", script), run_name="example")
    score = mod["main"](argv)
    assert score == score and score >= floor, f"{script}: score {score}"


def test_sctag_example_scalable_decoder(cuda_device):
    mod = runpy.run_path(os.path.join(ROOT, "examples", "single_modality/clustering/sctag.py"), run_name="example")
    assert mod["main"](["--cells", "1000", "--genes", "600", "--n_top_genes", "300", "--epochs", "6", "--pretrain_epochs", "10", "--adj_dim", "32"]) > 0.2
