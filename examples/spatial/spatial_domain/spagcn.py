"""examples/spatial/spatial_domain/spagcn.py of the reference, on a synthetic grid of spots: the pipeline (gene-name filter, normalisation,
the two spatial graphs, PCA) -> search_l -> search_set_res -> fit_predict -> ARI."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _synthetic import as_data, spots  # noqa: E402

from dance_amd.modules.spatial.spatial_domain.spagcn import SpaGCN, refine  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--side", type=int, default=40, help="spots per side of the grid")
    p.add_argument("--genes", type=int, default=600)
    p.add_argument("--n_clusters", type=int, default=4)
    p.add_argument("--beta", type=int, default=49)
    p.add_argument("--alpha", type=int, default=1)
    p.add_argument("--p", type=float, default=0.5)
    p.add_argument("--epochs", type=int, default=40)
    p.add_argument("--lr", type=float, default=0.05)
    p.add_argument("--tol", type=float, default=5e-3)
    p.add_argument("--max_run", type=int, default=6)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=100)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    x, domain, xy, xy_pixel, image = spots(args.side, args.n_clusters, args.genes, args.seed)
    names = [f"G{i}" for i in range(args.genes)]
    names[0], names[1] = "ERCC-00002", "MT-ND1"
    data = as_data(x, device=args.device, obs={"label": domain}, obsm={"spatial": xy, "spatial_pixel": xy_pixel}, uns={"image": image}, var_names=names)
    model = SpaGCN(device=args.device)
    model.preprocessing_pipeline(alpha=args.alpha, beta=args.beta)(data)
    (xf, adj, adj_2d), y = data.get_train_data()
    l = model.search_l(args.p, adj, start=0.01, end=1000, tol=0.01, max_run=100)
    model.set_l(l)
    res = model.search_set_res((xf, adj), l=l, target_num=args.n_clusters, start=0.4, step=0.1, tol=args.tol, lr=args.lr, epochs=args.epochs,
                               max_run=args.max_run)
    pred = model.fit_predict((xf, adj), init_spa=True, init="louvain", tol=args.tol, lr=args.lr, epochs=args.epochs, res=res)
    y = np.asarray(y).ravel()
    score = model.default_score_func(y, pred)
    refined = refine(sample_id=data.data.obs.index.tolist(), pred=pred.tolist(), dis=adj_2d, shape="square")
    score_refined = model.default_score_func(y, refined)
    print(f"SpaGCN l = {l:.4f}, res = {res}, ARI: {score:.4f}, refined: {score_refined:.4f}")
    return score_refined


if __name__ == "__main__":
    main()
