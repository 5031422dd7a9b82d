"""Oracle-backed compute namespace (torch-CPU / scipy) with the signature of ``dance_amd.kernels`` — TEST ONLY.

Injected into ``dance_amd.sharding.sharded_gcn_layer(ops=...)`` so that the partition + collective logic can be
exercised under gloo on a CPU box.  The product never selects this by itself."""
import numpy as np
import scipy.sparse as sp
import torch

ACT_NONE, ACT_RELU = 0, 1
REDUCE_SUM, REDUCE_MEAN = 0, 1


def gemm(A, B, *, trans_a=False, trans_b=False, out=None, accumulate=False, tag=None):
    r = torch.mm(A.t() if trans_a else A, B.t() if trans_b else B)
    if out is not None:
        out.copy_(out + r if accumulate else r)
        return out
    return r


def spmm_csr(rowptr, col, val, Z, *, n_cols=None, rowscale=None, colscale=None, bias=None, act=ACT_NONE,
             reduce=REDUCE_SUM, out=None, tag=None):
    n_rows = rowptr.numel() - 1
    n_cols = Z.shape[0] if n_cols is None else n_cols
    v = np.ones(col.numel(), np.float32) if val is None else val.numpy()
    a = sp.csr_matrix((v, col.numpy(), rowptr.numpy()), shape=(n_rows, n_cols))
    z = Z.numpy()
    if colscale is not None:
        z = z * colscale.numpy()[:, None]
    y = a @ z
    if reduce == REDUCE_MEAN:
        deg = np.diff(rowptr.numpy())
        y = y / np.maximum(deg, 1)[:, None]
    if rowscale is not None:
        y = y * rowscale.numpy()[:, None]
    if bias is not None:
        y = y + bias.detach().numpy()
    if act == ACT_RELU:
        y = np.maximum(y, 0)
    return torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))


def relu_backward(Y, dY):
    return torch.where(Y > 0, dY, torch.zeros_like(dY))


def colsum(X):
    return X.sum(0)


def knn(X, k, q_begin=0, q_end=None, *, algo=0):
    from oracle import graphs as og
    idx, dist = og.knn_exact(X.numpy(), k, q_begin=q_begin, q_end=X.shape[0] if q_end is None else q_end)
    return torch.from_numpy(idx), torch.from_numpy(dist)


def block_build(rowptr, col, val, seeds, mark, lut):
    """Torch-op restatement of ``dance_amd.kernels.block_build`` (dh_block_plan / dh_block_fill): all in-edges of ``seeds``;
    source nodes = seeds first, then the remaining in-neighbours by ascending id.  TEST ONLY (oracle for the HIP builder and
    stand-in for the loader logic tests on CPU tensors)."""
    seeds = seeds.to(torch.int64)
    n_nodes = rowptr.numel() - 1
    start = rowptr[seeds].to(torch.int64)
    deg = rowptr[seeds + 1].to(torch.int64) - start
    brp = torch.zeros(seeds.numel() + 1, dtype=torch.int64, device=rowptr.device)
    brp[1:] = torch.cumsum(deg, 0)
    total = int(brp[-1])
    pos = torch.repeat_interleave(start - brp[:-1], deg) + torch.arange(total, device=rowptr.device)
    gcol = col[pos].to(torch.int64)
    m = torch.zeros(n_nodes, dtype=torch.bool, device=rowptr.device)
    m[gcol] = True
    m[seeds] = False
    others = torch.nonzero(m).reshape(-1)
    src_ids = torch.cat((seeds, others))
    table = torch.empty(n_nodes, dtype=torch.int64, device=rowptr.device)
    table[src_ids] = torch.arange(src_ids.numel(), device=rowptr.device)
    return brp.to(torch.int32), table[gcol].to(torch.int32), None if val is None else val[pos].contiguous(), src_ids
