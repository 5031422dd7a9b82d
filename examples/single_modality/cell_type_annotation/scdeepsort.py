"""examples/single_modality/cell_type_annotation/scdeepsort.py of the reference, on synthetic cells: PCACellFeatureGraph -> ScDeepSort.fit
-> score on held-out cells.   python examples/single_modality/cell_type_annotation/scdeepsort.py --n_epochs 20"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from _synthetic import as_data, counts  # noqa: E402

from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cells", type=int, default=6000)
    p.add_argument("--genes", type=int, default=800)
    p.add_argument("--types", type=int, default=6)
    p.add_argument("--batch_size", type=int, default=500)
    p.add_argument("--dense_dim", type=int, default=400, help="number of PCA components")
    p.add_argument("--hidden_dim", type=int, default=200)
    p.add_argument("--n_layers", type=int, default=1)
    p.add_argument("--n_epochs", type=int, default=30)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--test_rate", type=float, default=0.2)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    x, types = counts(args.cells, args.genes, args.types, args.seed)
    n_train = int(args.cells * (1 - args.test_rate))
    data = as_data(x, device=args.device, obsm={"cell_type": np.eye(args.types, dtype=np.float32)[types]}, train_size=n_train)
    data.set_config(feature_channel=None, feature_channel_type="X")
    ScDeepSort.preprocessing_pipeline(n_components=min(args.dense_dim, args.genes - 1))(data)
    g = data.data.uns["CellFeatureGraph"]
    genes = torch.arange(args.genes)
    g_train = g.subgraph(torch.cat((genes, args.genes + torch.arange(n_train))))
    g_test = g.subgraph(torch.cat((genes, args.genes + torch.arange(n_train, args.cells))))
    with tempfile.TemporaryDirectory() as tmp:
        model = ScDeepSort(g.ndata["features"].shape[1], args.hidden_dim, args.n_layers, "synthetic", "example", batch_size=args.batch_size,
                           device=args.device, save_root=tmp, verbose=False)
        model.fit(g_train, torch.from_numpy(types[:n_train]), epochs=args.n_epochs, lr=args.lr, val_ratio=0.2)
        acc = model.score(g_test, np.eye(args.types, dtype=np.float32)[types[n_train:]])
    print(f"ScDeepSort ACC on {args.cells - n_train} held-out cells: {acc:.4f}")
    return acc


if __name__ == "__main__":
    main()
