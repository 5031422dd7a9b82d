#!/usr/bin/env python
"""development: time of dh_graphsc_steps phase 3 (the large-batch aggregation) at batch 8192 on the 1M-cell graph, gather vs matrix cores."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import _cellgene_graph
from dance_amd.ministep import GraphSCStepper
from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
cg = _cellgene_graph(n, 2000, 200, 50, dev)
gs = GraphSC(in_feats=50, n_clusters=10, device="cuda"); gs.model.train()
opt = torch.optim.Adam(gs.model.parameters(), lr=1e-5, fused=True)
st = GraphSCStepper(gs.model, cg, 8192, opt)
seeds = (2000 + torch.randperm(n, device=dev))[:8192].contiguous()
for _ in range(3): st.aggregate(seeds)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): st.aggregate(seeds)
b.record(); torch.cuda.synchronize()
print(os.environ.get("DANCE_AMD_MINISTEP_DBG"), os.environ.get("DANCE_AMD_GRAPHSC_AGG"), "ms per aggregate:", a.elapsed_time(b) / 20)
