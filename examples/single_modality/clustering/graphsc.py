"""examples/single_modality/clustering/graphsc.py of the reference, on synthetic cells: the device preprocessing pipeline -> GraphSC.fit
(mini-batches of 128 cells, one captured step per batch) -> k-means on the embedding -> ARI."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _synthetic import as_data, counts  # noqa: E402

from dance_amd.modules.single_modality.clustering.graphsc import GraphSC  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cells", type=int, default=10000)
    p.add_argument("--genes", type=int, default=1500)
    p.add_argument("--types", type=int, default=6)
    p.add_argument("--epochs", type=int, default=4)
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--nb_genes", type=int, default=1000)
    p.add_argument("--in_feats", type=int, default=50)
    p.add_argument("--learning_rate", type=float, default=1e-5)
    p.add_argument("--normalize_weights", default="log_per_cell", choices=["log_per_cell", "per_cell", "none"])
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=0)
    args = p.parse_args(argv)
    torch.manual_seed(args.seed)
    x, types = counts(args.cells, args.genes, args.types, args.seed)
    data = as_data(x, device=args.device, obs={"Group": types})
    GraphSC.preprocessing_pipeline(n_top_genes=args.nb_genes, normalize_weights=args.normalize_weights, n_components=args.in_feats)(data)
    graph = data.data.uns["CellFeatureGraph"]
    y = data.data.obs["Group"].to_numpy()
    model = GraphSC(in_feats=args.in_feats, n_clusters=args.types, device=args.device)
    model.fit(graph, epochs=args.epochs, lr=args.learning_rate, batch_size=args.batch_size)
    score = model.score(None, y)
    print(f"GraphSC ARI: {score:.4f} (loss {model.losses[0]:.4f} -> {model.losses[-1]:.4f})")
    return score


if __name__ == "__main__":
    main()
