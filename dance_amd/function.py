"""The slice of ``dgl.function`` the reference's layers hand to ``update_all`` (dance/models/nn/gnn.py:90,
dance/modules/single_modality/clustering/graphsc.py:463-465): built-in message / reduce descriptors.

``update_all(fn.u_mul_e("h", "weight", "m"), fn.sum("m", "h"))`` and ``update_all(fn.copy_u("h", "m"), fn.mean("m", "neigh"))`` map
one to one onto ``dh_spmm_csr_f32`` (``dance_amd.cellgraph``); a Python message UDF goes through an edge batch instead."""


class _Message:

    def __init__(self, kind: str, lhs: str, rhs, out: str):
        self.kind, self.lhs, self.rhs, self.out = kind, lhs, rhs, out


class _Reduce:

    def __init__(self, kind: str, msg: str, out: str):
        self.kind, self.msg, self.out = kind, msg, out


def u_mul_e(lhs_field: str, rhs_field: str, out: str) -> _Message:
    """message = source feature * edge feature (the edge feature is a scalar per edge: shape [E] or [E, 1])."""
    return _Message("u_mul_e", lhs_field, rhs_field, out)


def copy_u(u: str, out: str) -> _Message:
    return _Message("copy_u", u, None, out)


def sum(msg: str, out: str) -> _Reduce:  # noqa: A001 (dgl.function's own name)
    return _Reduce("sum", msg, out)


def mean(msg: str, out: str) -> _Reduce:
    """Mean over the in-edges; 0 for a destination without in-edges (DGL's convention)."""
    return _Reduce("mean", msg, out)
