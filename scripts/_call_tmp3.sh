R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v14; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_sage_dense.py -m gpu -x -q -k mfma > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python scripts/sage_mfma_bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
DANCE_HIP_LIB=$R/build/variants/libdancehip_smprof.so python - <<'PY'
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from dance_amd import kernels, _lib
lib = _lib.load()
lib.dh_sage_mfma_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev="cuda"; n_cells=200_000; n_genes, dfeat, per = 2000, 400, 200
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32)
col = torch.cat((col, (n_genes + torch.arange(n_cells, device=dev, dtype=torch.int32))[:, None]), 1).reshape(-1).contiguous()
rowptr = torch.arange(0, n_cells * (per + 1) + 1, per + 1, dtype=torch.int32, device=dev)
w = torch.rand(col.numel(), device=dev, generator=g) + 0.5
feats = torch.randn(n_genes + n_cells, dfeat, device=dev, generator=g)
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5
args = (rowptr, col, w, cid, cid[n_genes:].contiguous(), alpha, feats)
kernels.sage_aggregate_mfma(*args, 0, n_genes); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.dh_sage_mfma_prof_read(buf, 1)
kernels.sage_aggregate_mfma(*args, 0, n_genes); torch.cuda.synchronize()
lib.dh_sage_mfma_prof_read(buf, 0)
names = ["prologue", "zero+barrier", "scatter", "barrier+fetch", "dma issue+frag reads", "mfma", "wait+barrier", "epilogue"]
nb = (n_cells + 127) // 128
tot = sum(buf)
for n, v in zip(names, buf): print(f"{n:24s} {v / nb:10.0f} cycles/block  {100 * v / tot:5.1f} %")
print("total cycles/block", tot / nb)
PY
