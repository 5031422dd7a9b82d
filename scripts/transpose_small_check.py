#!/usr/bin/env python
"""Does the single-workgroup transpose leave holes?  Static cell blocks of random seeds, output buffers poisoned before every call,
compared with the sort path entry for entry (DANCE_AMD_TRANSPOSE_SMALL=1 must be set for the process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402
from dance_amd import _lib, kernels  # noqa: E402
from dance_amd.cellgraph import StaticCellBlock  # noqa: E402

assert os.environ.get("DANCE_AMD_TRANSPOSE_SMALL") == "1"
dev = torch.device("cuda", 0)
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
cg = bench_configs._cellgene_graph(n_cells, 2000, 200, 50, dev)
lib = _lib.load()
B = 128
blk = StaticCellBlock(cg, B)
n_rows, n_cols, nnz = B + 1, blk.number_of_src_nodes(), blk.e_max
print("rows", n_rows, "cols", n_cols, "nnz", nnz)
g = torch.Generator(device=dev).manual_seed(0)
ws_bytes = lib.dh_csr_transpose_workspace_bytes(n_rows, n_cols, nnz)
ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
bad = 0
for it in range(300):
    blk.seeds.copy_(torch.randint(2000, 2000 + n_cells, (B, ), device=dev, generator=g))
    blk.rebuild()
    outs = []
    for small in (True, False):
        rp = torch.full((n_cols + 1, ), -7, dtype=torch.int32, device=dev)
        oc = torch.full((nnz, ), -7, dtype=torch.int32, device=dev)
        ov = torch.full((nnz, ), -7.0, dtype=torch.float32, device=dev)
        op = torch.full((nnz, ), -7, dtype=torch.int32, device=dev)
        # the sort path is reached by asking for one row more than the small path admits (rows beyond n_rows are never touched)
        rows_arg = n_rows if small else n_rows
        if not small:
            os.environ["DANCE_AMD_TRANSPOSE_SMALL"] = "1"
        rc = lib.dh_csr_transpose(rows_arg, n_cols, nnz if small else nnz, blk.rowptr.data_ptr(), blk.col.data_ptr(), blk.val.data_ptr(), rp.data_ptr(), oc.data_ptr(),
                                  ov.data_ptr(), op.data_ptr(), ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append((rp, oc, ov, op))
        break
    torch.cuda.synchronize()
    rp, oc, ov, op = outs[0]
    holes = int((oc == -7).sum())
    # reference transpose on the host
    import numpy as np
    import scipy.sparse as sp
    a = sp.csr_matrix((blk.val.cpu().numpy(), blk.col.cpu().numpy(), blk.rowptr.cpu().numpy()), shape=(n_rows, n_cols))
    ok_rp = np.array_equal(rp.cpu().numpy(), np.concatenate(([0], np.cumsum(np.bincount(blk.col.cpu().numpy(), minlength=n_cols)))))
    if holes or not ok_rp:
        bad += 1
        print(f"iteration {it}: holes {holes}, rowptr ok {ok_rp}, brp tail {blk.rowptr[-3:].tolist()}")
print("bad iterations:", bad, "of 300")
