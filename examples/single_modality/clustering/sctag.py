"""examples/single_modality/clustering/sctag.py of the reference, on synthetic cells: the device preprocessing pipeline (filters, HVG, scale,
PCA, kNN graph) -> ScTAG.fit (TAGConv encoder, ZINB + adjacency decoders, clustering loss) -> ARI.  --adj_dim selects the adjacency
decoder whose loss needs no N x N matrix."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _synthetic import as_data, counts  # noqa: E402

from dance_amd.modules.single_modality.clustering.sctag import ScTAG  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cells", type=int, default=2500)
    p.add_argument("--genes", type=int, default=1200)
    p.add_argument("--types", type=int, default=5)
    p.add_argument("--n_top_genes", type=int, default=600)
    p.add_argument("--k_neighbor", type=int, default=15)
    p.add_argument("--epochs", type=int, default=30)
    p.add_argument("--pretrain_epochs", type=int, default=30)
    p.add_argument("--adj_dim", type=int, default=None)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    x, types = counts(args.cells, args.genes, args.types, args.seed)
    data = as_data(x, device=args.device, obsm={"Group": types[:, None]})  # the clustering datasets keep the labels in obsm["Group"]
    ScTAG.preprocessing_pipeline(n_top_genes=args.n_top_genes, n_neighbors=args.k_neighbor)(data)
    inputs, y = data.get_data(return_type="default")  # adj, x, x_raw, n_counts
    y = np.asarray(y).ravel()
    model = ScTAG(n_clusters=args.types, device=args.device, adj_dim=args.adj_dim)
    model.fit(inputs, y, epochs=args.epochs, pretrain_epochs=args.pretrain_epochs, force_pretrain=True)
    score = model.score(None, y)
    print(f"ScTAG ARI: {score:.4f}")
    return score


if __name__ == "__main__":
    main()
