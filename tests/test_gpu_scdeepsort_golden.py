"""GPU: ScDeepSort.fit / cal_loss / evaluate / predict_proba / predict through the HIP kernels against
tests/golden/scdeepsort.npz — the numbers the reference's OWN loop (scdeepsort.py:142-349, AST-lifted with its GNN and
AdaptiveSAGE, run on torch-CPU over the DGL stub) produced: the train / validation split, per-epoch loss, the
(correct, unsure, accuracy) triples with the raw-logit "unsure" rule (:280-281), the best-validation checkpoint, the class
probabilities and predict's flags.  Body shared with the CPU host-logic twin: tests/scdeepsort_golden_checks.py."""
import pytest

import scdeepsort_golden_checks as chk
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,block_eval", [("one", True), ("mb", True), ("peak", True), ("one", False), ("peak", False)])
def test_scdeepsort_fit_loop_vs_reference(cuda_device, tmp_path, monkeypatch, tag, block_eval):
    """block_eval=True: evaluation batch by batch over sampled blocks, as the reference does (and consuming the loader's
    permutations in the reference's order).  block_eval=False: the product's default, one pass over the CSR rows of all cells —
    "one" must land on the same numbers (one batch per epoch: the seed order only permutes a sum); "peak" runs with different
    batches after the first epoch (the evaluation loaders no longer draw permutations), so only its first epoch is compared."""
    gold, kw = chk.load()
    if tag == "peak" and not block_eval:
        m, g, log = chk.fit_case(gold, kw, tag, "cuda", tmp_path, monkeypatch, block_eval=False)
        assert abs(log["cal_loss"][0] - gold["peak_losses"][0]) < 2e-4 * gold["peak_losses"][0]
        assert [tuple(v) for v in log["evaluate"][:2]] == [tuple(v) for v in gold["peak_eval"][:2].tolist()]
        return
    chk.check_case(gold, kw, tag, "cuda", tmp_path, monkeypatch, block_eval=block_eval, rel_err=rel_err)
