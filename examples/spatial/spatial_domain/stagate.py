"""examples/spatial/spatial_domain/stagate.py of the reference, on a synthetic grid of spots: HVG (dispersion flavour) / normalisation / the
kNN spot graph on the device -> Stagate.fit (tied-weight graph-attention auto-encoder + Gaussian mixture) -> ARI."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _synthetic import as_data, spots  # noqa: E402

from dance_amd.modules.spatial.spatial_domain.stagate import Stagate  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--side", type=int, default=40)
    p.add_argument("--genes", type=int, default=800)
    p.add_argument("--n_clusters", type=int, default=4)
    p.add_argument("--hidden_dims", type=int, nargs=2, default=[128, 30])
    p.add_argument("--high_variable_genes", type=int, default=400)
    p.add_argument("--hvg_flavor", default="seurat_v3")
    p.add_argument("--n_neighbors", type=int, default=6)
    p.add_argument("--epochs", type=int, default=60)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=3)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    x, domain, xy, xy_pixel, _ = spots(args.side, args.n_clusters, args.genes, args.seed)
    data = as_data(x, device=args.device, obs={"label": domain}, obsm={"spatial": xy, "spatial_pixel": xy_pixel})
    Stagate.preprocessing_pipeline(hvg_flavor=args.hvg_flavor, n_top_hvgs=args.high_variable_genes, model_name="knn",
                                   n_neighbors=args.n_neighbors)(data)
    adj, y = data.get_data(return_type="default")
    xm = np.asarray(data.data.X, dtype=np.float32)
    adj = adj.materialize() if hasattr(adj, "materialize") else adj
    edge_list = np.vstack(np.nonzero(adj))
    model = Stagate([xm.shape[1]] + list(args.hidden_dims), device=args.device)
    score = model.fit_score((xm, edge_list), np.asarray(y).ravel(), epochs=args.epochs, num_cluster=args.n_clusters, random_state=args.seed)
    print(f"STAGATE ARI: {score:.4f}")
    return score


if __name__ == "__main__":
    main()
