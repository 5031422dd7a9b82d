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
             reduce=REDUCE_SUM, out=None, rows=None, tag=None):
    y = _spmm_full(rowptr, col, val, Z, n_cols, rowscale, colscale, bias, act, reduce)
    if out is None:
        if rows is not None:
            raise ValueError("rows needs out")
        return y
    if rows is None:
        out.copy_(y)
    else:
        out[rows.long()] = y[rows.long()]
    return out


def _spmm_full(rowptr, col, val, Z, n_cols, rowscale, colscale, bias, act, reduce):
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


# ---- fused ReLU sign masks, in the byte layout of dh_spmm_csr_relu_f32: per row and 128-column slice four 32-bit words
#      = a little-endian 128-bit bitmap: bit c % 32 of word c / 32 = [element at column slice * 128 + c is > 0] ---------------
def relu_mask_bytes(n_rows, width):
    return 0 if (n_rows <= 0 or width <= 0 or width % 128) else n_rows * (width // 128) * 16


def _mask_to_bool(mask, n_rows, width):
    words = mask.numpy()[:n_rows * (width // 128) * 16].view(np.uint32).reshape(n_rows, width // 32)
    bits = (words[..., None] >> np.arange(32, dtype=np.uint32)) & 1          # [row, word, bit]
    return bits.reshape(n_rows, width).astype(bool)


def _bool_to_mask(b):
    n_rows, width = b.shape
    bits = b.reshape(n_rows, width // 32, 32).astype(np.uint32)
    words = (bits << np.arange(32, dtype=np.uint32)).sum(-1, dtype=np.uint64).astype(np.uint32)
    return torch.from_numpy(words.reshape(-1).view(np.uint8).copy())


def spmm_csr_relu(rowptr, col, val, Z, *, n_cols=None, bias=None, act=ACT_NONE, out_mask=None, in_mask=None, out=None, rows=None,
                  tag=None):
    n_rows, width = rowptr.numel() - 1, Z.shape[1]
    if in_mask is not None:
        Z = torch.where(torch.from_numpy(_mask_to_bool(in_mask, Z.shape[0], width)), Z, torch.zeros_like(Z))
    y = _spmm_full(rowptr, col, val, Z, n_cols, None, None, bias, act, REDUCE_SUM)
    sel = slice(None) if rows is None else rows.long()
    if out is None:
        out = torch.empty_like(y)
    out[sel] = y[sel]
    if out_mask is not None:
        m = _bool_to_mask((y > 0).numpy()).reshape(n_rows, -1)
        view = out_mask[:n_rows * m.shape[1]].reshape(n_rows, -1)
        view[sel] = m[sel]
    return out


def gather_rows(X, idx, *, relu_mask=None, out=None):
    r = X[idx.long()]
    if relu_mask is not None:
        r = torch.where(torch.from_numpy(_mask_to_bool(relu_mask, X.shape[0], X.shape[1]))[idx.long()], r, torch.zeros_like(r))
    if out is not None:
        out.copy_(r)
        return out
    return r.contiguous()


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


def csr_transpose(rowptr, col, val, n_rows, n_cols):
    """(rowptr_t, col_t, val_t, perm) of A^T, stable by input position (dh_csr_transpose)."""
    nnz = col.numel()
    rows = torch.repeat_interleave(torch.arange(n_rows), (rowptr[1:] - rowptr[:-1]).to(torch.int64))
    perm = torch.sort(col.to(torch.int64), stable=True).indices
    counts = torch.bincount(col.to(torch.int64), minlength=n_cols)
    rowptr_t = torch.zeros(n_cols + 1, dtype=torch.int64)
    rowptr_t[1:] = torch.cumsum(counts, 0)
    assert perm.numel() == nnz
    return rowptr_t.to(torch.int32), rows[perm].to(torch.int32), None if val is None else val[perm].contiguous(), perm.to(torch.int32)


def bias_act_(X, bias, act=ACT_NONE):
    if bias is not None:
        X.add_(bias.detach())
    if act == ACT_RELU:
        X.clamp_(min=0)
    return X


def softplus_rowsum(X):
    return torch.nn.functional.softplus(X.double()).sum(1).float()


def sigmoid_scale(X, scale):
    return torch.sigmoid(X) * scale.reshape(())


def gram_sigmoid_supported(n, d):
    return 1 <= d <= 320


def gram_sigmoid(Z):
    x = Z.double() @ Z.double().t()
    return torch.nn.functional.softplus(x).sum(1).float(), (torch.sigmoid(x) @ Z.double()).float()
