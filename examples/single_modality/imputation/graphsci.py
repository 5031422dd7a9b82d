"""examples/single_modality/imputation/graphsci.py of the reference, on synthetic cells: the device preprocessing pipeline (filters,
log1p, top-variance genes, gene-gene correlation graph, entry masks) -> GraphSCI.fit -> the held-out entries imputed.  Returns
1 - RMSE(imputed) / RMSE(zeros) over the test-masked entries of the test cells: positive when imputing beats leaving the dropouts."""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _synthetic import as_data, counts  # noqa: E402

from dance_amd.modules.single_modality.imputation.graphsci import GraphSCI  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cells", type=int, default=1500)
    p.add_argument("--genes", type=int, default=800)
    p.add_argument("--types", type=int, default=5)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--train_size", type=float, default=0.9)
    p.add_argument("--le", type=float, default=1)
    p.add_argument("--la", type=float, default=1e-9)
    p.add_argument("--ke", type=float, default=1e2)
    p.add_argument("--ka", type=float, default=1)
    p.add_argument("--n_epochs", type=int, default=100)
    p.add_argument("--weight_decay", type=float, default=1e-6)
    p.add_argument("--threshold", type=float, default=.3)
    p.add_argument("--mask_rate", type=float, default=.1)
    p.add_argument("--min_cells", type=float, default=.05)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    x, _ = counts(args.cells, args.genes, args.types, args.seed)
    data = as_data(x, device=args.device, train_size=int(args.cells * args.train_size))
    pipe = GraphSCI.preprocessing_pipeline(min_cells=args.min_cells, threshold=args.threshold, mask=True, seed=args.seed, mask_rate=args.mask_rate)
    for t in pipe.transforms:
        if hasattr(t, "device"):
            t.device = args.device
    pipe(data)
    X, X_raw, g, mask, valid_mask, test_mask = data.get_x(return_type="default")
    dev = torch.device(args.device)
    X = torch.as_tensor(np.asarray(X), dtype=torch.float32).to(dev)       # DeviceArray slots: materialised once for the masks below
    X_raw = torch.as_tensor(np.asarray(X_raw), dtype=torch.float32).to(dev)
    m = torch.from_numpy(mask).to(dev)
    X_train, X_raw_train = X * m, X_raw * m
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)  # the model keeps its best-validation checkpoint under ./graphsci/, as the reference does
        try:
            model = GraphSCI(num_cells=X.shape[0], num_genes=X.shape[1], dataset="synthetic", dropout=args.dropout, seed=args.seed, device=dev)
            model.fit(X_train, X_raw_train, g, mask, args.le, args.la, args.ke, args.ka, args.n_epochs, args.lr, args.weight_decay,
                      train_idx=data.train_idx)
            model.load_model()
            imputed = model.predict(X_train, X_raw_train, g, mask)
        finally:
            os.chdir(cwd)
    test_rmse = model.score(X, imputed.clone(), ~test_mask, "RMSE", log1p=False, test_idx=data.test_idx)
    zero_rmse = model.score(X, torch.zeros_like(imputed), ~test_mask, "RMSE", log1p=False, test_idx=data.test_idx)
    test_mre = model.score(X, imputed.clone(), ~test_mask, "MRE", log1p=False, test_idx=data.test_idx)
    print(f"GraphSCI test RMSE: {test_rmse:.4f} (zeros: {zero_rmse:.4f}), MRE: {test_mre:.4f}")
    return 1.0 - test_rmse / zero_rmse


if __name__ == "__main__":
    main()
