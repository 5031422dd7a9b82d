"""examples/single_modality/cell_type_annotation/scheteronet.py of the reference, on synthetic cells: the device preprocessing pipeline
(rare types, filters, HVG, size factors, kNN graph) -> set_split / convert_dgl_to_original_format -> scHeteroNet.fit per epoch (NLL +
ZINB + contrastive terms) -> accuracy on the held-out in-distribution cells."""
import argparse
import os
import sys

import numpy as np
import pandas as pd
import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from _synthetic import as_data, counts  # noqa: E402

from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import (  # noqa: E402
    convert_dgl_to_original_format, scHeteroNet, set_split)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cells", type=int, default=4000)
    p.add_argument("--genes", type=int, default=1500)
    p.add_argument("--types", type=int, default=6)
    p.add_argument("--hidden_channels", type=int, default=64)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--epochs", type=int, default=30)
    p.add_argument("--use_zinb", action="store_true")
    p.add_argument("--zinb_weight", type=float, default=1e-4)
    p.add_argument("--cl_weight", type=float, default=0.0)
    p.add_argument("--mask_ratio", type=float, default=0.8)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=42)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    x, types = counts(args.cells, args.genes, args.types, args.seed)
    types[:60] = args.types - 1   # make the last type the rarest but above FilterCellsType's threshold: it becomes the OOD class
    types[60:][types[60:] == args.types - 1] = 0
    one_hot = pd.DataFrame(np.eye(args.types, dtype=np.float32)[types], columns=[f"type{i}" for i in range(args.types)],
                           index=[str(i) for i in range(args.cells)])
    n_train, n_val = int(0.6 * args.cells), int(0.2 * args.cells)
    data = as_data(x, device=args.device, obsm={"cell_type": one_hot}, train_size=n_train, val_size=n_val, test_size=args.cells - n_train - n_val)
    scHeteroNet.preprocessing_pipeline()(data)
    set_split(data, data.train_idx, data.val_idx, data.test_idx)
    g = data.data.uns["HeteronetGraph"]
    dataset_ind, dataset_ood_tr, dataset_ood_te, adata = convert_dgl_to_original_format(g, data.data, "synthetic")
    for ds in (dataset_ind, dataset_ood_tr, dataset_ood_te):
        if ds.y.dim() == 1:
            ds.y = ds.y.unsqueeze(1)
    c = max(int(dataset_ind.y.max()) + 1, dataset_ind.y.shape[1])
    d = dataset_ind.graph["node_feat"].shape[1]
    model = scHeteroNet(d, c, dataset_ind.edge_index.to(args.device), dataset_ind.num_nodes, hidden_channels=args.hidden_channels,
                        num_layers=args.num_layers, dropout=args.dropout, use_bn=False, device=args.device, min_loss=100000)
    criterion = nn.NLLLoss()
    model.train()
    model.reset_parameters()
    model.to(args.device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    test_idx = adata.uns["test_idx"]
    for epoch in range(args.epochs):
        loss = model.fit(dataset_ind, dataset_ood_tr, args.use_zinb, adata, args.zinb_weight, args.cl_weight, args.mask_ratio, criterion, optimizer)
    score = model.score(dataset_ind, dataset_ind.y, test_idx)
    print(f"scHeteroNet loss {float(loss):.4f}, test score (in-distribution cells): {score:.4f}")
    return float(score)


if __name__ == "__main__":
    main()
