"""Entry masking for the imputation models (dance/transforms/mask.py:79-290): per cell, a fraction of the expressed entries is held
out of training.  The draw is a sequential numpy ``Generator`` stream — one ``choice`` (and, with ``add_test_mask``, one
``permutation``) per cell in cell order — so it stays on the host: the same seed gives the reference's masks bit for bit.  What
the models consume are the three boolean N x G layers."""
from typing import Optional

import numpy as np
import scipy.sparse as sp

from ..data import DeviceArray
from ..registry import register_preprocessor
from .base import BaseTransform


@register_preprocessor("split", "entry")
class CellwiseMaskData(BaseTransform):
    """Hold out ``floor(mask_rate * #expressed)`` expressed entries of every cell with more than ``min_gene_counts`` of them.
    ``distr="exp"`` draws them with probability proportional to the Exp(scale=20) density of the value (small counts go first),
    ``"uniform"`` evenly.  ``add_test_mask``: about a tenth of a cell's held-out entries (at least one) go to ``valid_mask``, the
    rest to ``test_mask``; otherwise all go to ``valid_mask``.  Writes ``layers["train_mask" | "valid_mask" | "test_mask"]``."""

    _DISPLAY_ATTRS = ("distr", "mask_rate", "seed", "min_gene_counts", "add_test_mask")

    def __init__(self, distr: Optional[str] = "exp", mask_rate: Optional[float] = 0.1, seed: Optional[int] = None,
                 min_gene_counts: int = 5, add_test_mask: bool = False, **kwargs):
        super().__init__(**kwargs)
        if not 0.0 <= mask_rate <= 1.0:
            raise ValueError(f"mask_rate must be between 0 and 1, got {mask_rate}")
        self.distr, self.mask_rate, self.seed = distr, mask_rate, seed
        self.min_gene_counts, self.add_test_mask = min_gene_counts, add_test_mask

    def _get_probs(self, vec):
        if self.distr == "exp":
            prob = np.exp(-np.asarray(vec, dtype=np.float64) / 20) / 20   # scipy.stats.expon.pdf(vec, 0, 20)
        elif self.distr == "uniform":
            prob = np.ones(len(vec))
        else:
            raise ValueError(f"Unknown distribution function option {self.distr!r}, "
                             "available options are: 'exp', 'uniform'")
        total = prob.sum()
        if total > 1e-9:
            return prob / total
        self.logger.warning("Probability sum is zero, falling back to uniform probability.")
        return np.ones(len(vec)) / len(vec) if len(vec) > 0 else np.array([])

    def __call__(self, data):
        rng = np.random.default_rng(self.seed)
        feat = data.get_feature(return_type="default", channel_type="X")
        if isinstance(feat, DeviceArray):   # the draw walks the rows on the host: one copy of the matrix, as a CSR
            feat = sp.csr_matrix(feat.numpy())
        if not sp.issparse(feat):
            feat = sp.csr_matrix(np.asarray(feat))
        feat = feat.tocsr()
        n_cells, n_genes = feat.shape
        train_mask = np.ones((n_cells, n_genes), dtype=bool)
        valid_mask = np.zeros((n_cells, n_genes), dtype=bool)
        test_mask = np.zeros((n_cells, n_genes), dtype=bool)
        for c in range(n_cells):
            lo, hi = feat.indptr[c], feat.indptr[c + 1]
            genes, values = feat.indices[lo:hi], feat.data[lo:hi]
            n_pos = hi - lo
            if n_pos <= self.min_gene_counts:
                continue
            n_masked = int(np.floor(n_pos * self.mask_rate))
            if n_masked <= 0:
                continue
            if n_masked >= n_pos:
                self.logger.warning(f"Mask rate {self.mask_rate} resulted in attempting to mask all "
                                    f"{n_pos} positive counts for cell {c}. Reducing mask count.")
                n_masked = 1 + int(np.floor(0.5 * n_pos))
            prob = self._get_probs(values) if self.distr == "exp" else None
            if prob is not None and (len(prob) != n_pos or not np.isclose(prob.sum(), 1.0)):
                self.logger.warning(f"Invalid probabilities calculated for cell {c}. Falling back to uniform.")
                prob = None
            try:
                held = genes[rng.choice(n_pos, size=n_masked, p=prob, replace=False)]
            except ValueError as e:
                self.logger.error(f"Error during rng.choice for cell {c}: {e}. Skipping masking for this cell.")
                continue
            train_mask[c, held] = False
            if self.add_test_mask:
                n_valid = max(1, int(np.round(len(held) * 0.1))) if len(held) > 1 else len(held)
                shuffled = rng.permutation(held)
                valid_mask[c, shuffled[:n_valid]] = True
                test_mask[c, shuffled[n_valid:]] = True
            else:
                valid_mask[c, held] = True
        data.data.layers["train_mask"] = train_mask
        data.data.layers["valid_mask"] = valid_mask
        data.data.layers["test_mask"] = test_mask
        n_total = n_cells * n_genes
        self.logger.info(f"Masking complete. Total elements: {n_total}; train {int(train_mask.sum())}, "
                         f"valid {int(valid_mask.sum())}, test {int(test_mask.sum())}")
        return data
