import sys, json, os
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import torch
import bench_configs as bc
from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
dev = torch.device("cuda", 0)
for flag in (True, False, True):
    ScDSC.cache_first_aggregation = flag
    r = bc.c2_scdsc_epoch(dev, 1_000_000, 1, 3, cpu_baseline={"skipped": True})
    print(flag, r["ms"], r["other_ms"], {k: v for k, v in r["kernels_ms"].items() if "spmm" in k or "gemm" in k}, flush=True)
    torch.cuda.empty_cache()
