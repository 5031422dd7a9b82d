"""Device-resident CSR adjacency for message passing (rows = destination nodes, columns = source nodes).

The reference hands its GCN layers a ``torch.sparse`` COO tensor built from a scipy matrix
(dance/transforms/preprocess.py:526-532, used at dance/modules/single_modality/clustering/scdsc.py:244) or a
dense FloatTensor (dance/modules/spatial/spatial_domain/spagcn.py:497).  ``CSRGraph`` is what our kernels
consume: int32 ``rowptr``/``col`` + f32 ``val`` in HBM, plus the lazily built CSR of the transpose used by
the backward SpMM (deterministic gather instead of atomic scatter).
"""
import weakref
from typing import Optional

import numpy as np
import torch

from . import kernels


class CSRGraph:

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], n_rows: int,
                 n_cols: int, *, symmetric: bool = False):
        if rowptr.dtype != torch.int32 or col.dtype != torch.int32:
            raise TypeError("CSRGraph wants int32 rowptr/col")
        if rowptr.numel() != n_rows + 1:
            raise ValueError(f"rowptr has {rowptr.numel()} entries, expected {n_rows + 1}")
        if val is not None and val.numel() != col.numel():
            raise ValueError("val and col differ in length")
        self.rowptr, self.col, self.val = rowptr, col, val
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.symmetric = bool(symmetric)  # value-symmetric square matrix: A^T == A, no transpose needed
        self._t: Optional["CSRGraph"] = None
        # rows whose entries the transpose is built from (None: all).  A StaticCellBlock's CSR ends with a padding row of zeros that
        # only exists to keep the entry count static: its transpose needs the real rows alone (every consumer goes by the row pointers)
        self.t_rows: Optional[int] = None

    # ---- constructors --------------------------------------------------------------------------------------
    @classmethod
    def from_scipy(cls, mat, device="cuda", *, symmetric: bool = False) -> "CSRGraph":
        """Host scipy sparse matrix -> device CSR (index order preserved after canonical CSR conversion)."""
        import scipy.sparse as sp
        if isinstance(mat, LazyScipyCSR):  # left in obsp by an on-device graph transform: the device graph is already there
            g = mat.graph
            if torch.device(device).type == g.device.type:
                return g
            mat = mat.materialize()
        csr = sp.csr_matrix(mat)
        csr.sum_duplicates()
        csr.sort_indices()
        if csr.nnz >= 2**31:
            raise ValueError("nnz >= 2^31 is not supported by the int32 CSR layout")
        rowptr = torch.from_numpy(csr.indptr.astype(np.int32)).to(device)
        col = torch.from_numpy(csr.indices.astype(np.int32)).to(device)
        val = torch.from_numpy(csr.data.astype(np.float32)).to(device)
        return cls(rowptr, col, val, csr.shape[0], csr.shape[1], symmetric=symmetric)

    @classmethod
    def from_torch_sparse(cls, adj: torch.Tensor, device=None) -> "CSRGraph":
        """torch sparse COO/CSR tensor (what the reference layers receive) -> device CSR.

        Duplicate COO entries are summed (``coalesce``), which is what ``torch.spmm`` computes.
        """
        if adj.layout == torch.sparse_coo:
            adj = adj.coalesce().to_sparse_csr()
        elif adj.layout != torch.sparse_csr:
            raise TypeError(f"unsupported adjacency layout {adj.layout}")
        device = device or (adj.device if adj.is_cuda else "cuda")
        rowptr = adj.crow_indices().to(torch.int32).to(device)
        col = adj.col_indices().to(torch.int32).to(device)
        val = adj.values().to(torch.float32).to(device)
        return cls(rowptr, col, val, adj.shape[0], adj.shape[1])

    # ---- views ---------------------------------------------------------------------------------------------
    @property
    def nnz(self) -> int:
        return int(self.col.numel())

    @property
    def device(self):
        return self.rowptr.device

    def transpose(self) -> "CSRGraph":
        """CSR of A^T, built once on the device (dh_csr_transpose) and cached."""
        if self.symmetric:
            return self
        if self._t is None:
            rp, c, v, _ = kernels.csr_transpose(self.rowptr, self.col, self.val, self.n_rows if self.t_rows is None else self.t_rows, self.n_cols)
            self._t = CSRGraph(rp, c, v, self.n_cols, self.n_rows)
            self._t._t = self
        return self._t

    def permute(self, perm: torch.Tensor) -> "CSRGraph":
        """P A P^T for ``perm[new] = old`` (square graphs): node ``perm[i]`` becomes node ``i``.

        Within every row the edges KEEP their order (ascending OLD column id), so each output element of an SpMM on the
        renumbered graph is the same fmaf chain as on the original: results are bit-identical after un-permutation.  The
        cached transpose is renumbered the same way (not rebuilt), so the backward sums keep their order as well.  Torch
        index ops on the device; set-up only."""
        if self.n_rows != self.n_cols:
            raise ValueError("permute needs a square graph")
        perm = perm.to(device=self.rowptr.device, dtype=torch.int64)
        n = self.n_rows
        if perm.numel() != n:
            raise ValueError(f"perm has {perm.numel()} entries for {n} nodes")
        inv = torch.empty(n, dtype=torch.int64, device=perm.device)
        inv[perm] = torch.arange(n, device=perm.device)

        def renumber(g):
            rp = g.rowptr.long()
            deg = (rp[1:] - rp[:-1])[perm]
            new_rp = torch.zeros(n + 1, dtype=torch.int64, device=perm.device)
            new_rp[1:] = torch.cumsum(deg, 0)
            nnz = int(new_rp[-1])
            src = torch.repeat_interleave(rp[:-1][perm] - new_rp[:-1], deg, output_size=nnz) + torch.arange(nnz, device=perm.device)
            col = inv[g.col.long()[src]].to(torch.int32)
            val = None if g.val is None else g.val[src].contiguous()
            return CSRGraph(new_rp.to(torch.int32), col.contiguous(), val, n, n, symmetric=g.symmetric)

        out = renumber(self)
        if self._t is not None and not self.symmetric:
            out._t = renumber(self._t)
            out._t._t = out
        return out

    def to_scipy(self):
        import scipy.sparse as sp
        val = self.val.cpu().numpy() if self.val is not None else np.ones(self.nnz, dtype=np.float32)
        return sp.csr_matrix((val, self.col.cpu().numpy(), self.rowptr.cpu().numpy()),
                             shape=(self.n_rows, self.n_cols))


class LazyScipyCSR:
    """What a graph transform leaves in ``obsp[...]`` next to the device graph: behaves as the scipy CSR matrix the reference
    stores there (attribute access, indexing, ``toarray`` ... are forwarded), but the device -> host copy happens on first use —
    a pipeline whose consumers take the device graph (``uns["<name>.hip"]``) never pays for it."""

    host_copies = 0

    def __init__(self, graph: "CSRGraph"):
        object.__setattr__(self, "graph", graph)
        object.__setattr__(self, "_m", None)

    def materialize(self):
        if self._m is None:
            object.__setattr__(self, "_m", self.graph.to_scipy())
            LazyScipyCSR.host_copies += 1
        return self._m

    @property
    def shape(self):
        return (self.graph.n_rows, self.graph.n_cols)

    @property
    def nnz(self):
        return self.graph.nnz

    def __getattr__(self, name):
        # only reached when normal lookup fails; private / dunder names are never forwarded (copy and pickle probe ``__deepcopy__``,
        # ``__getstate__``, ``_m`` ... on an instance whose __init__ has not run: forwarding those recursed)
        if name.startswith("_") or name == "graph":
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __reduce__(self):
        return (LazyScipyCSR, (self.graph, ))

    def __deepcopy__(self, memo):
        import copy
        return LazyScipyCSR(copy.deepcopy(self.graph, memo))

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __repr__(self):
        return f"LazyScipyCSR(shape={self.shape}, nnz={self.nnz}, materialised={self._m is not None})"


def morton_order(coords: torch.Tensor, dims: int = 3) -> torch.Tensor:
    """``perm[new] = old`` along a Z-order curve through the leading ``dims`` principal components of ``coords`` [N, d] — entirely
    on the device the coordinates live on: covariance (d x d) -> eigenvectors -> projection -> 21-bit quantisation per axis ->
    bit interleave -> one sort.  A kNN graph is BUILT from such coordinates (NeighborGraph's ``channel``), so rows that are close
    on the curve are close in the space the neighbours were searched in.  Deterministic; ties keep the input order."""
    if coords.dim() != 2 or not 1 <= dims <= 3:
        raise ValueError("morton_order: coords must be [N, d] and dims in 1..3")
    x = coords.to(torch.float32)
    n, d = x.shape
    mean = x.mean(0, keepdim=True)
    xc = x - mean
    on_gpu = x.is_cuda
    if on_gpu:  # the library's own products: torch's `@` brings the vendor BLAS up on its first call (tens of ms of a 90 ms set-up step)
        from . import kernels
        xc = xc.contiguous()
        cov = kernels.gemm(xc, xc, trans_a=True).double() / max(n - 1, 1)
    else:
        cov = (xc.t() @ xc).double() / max(n - 1, 1)                   # d x d: the one reduction over all rows
    # the d x d eigen-decomposition on the host (d = 50: microseconds; the device solver's first call alone costs ~0.15 s)
    evals, evecs = np.linalg.eigh(cov.cpu().numpy())
    axes = torch.from_numpy(evecs[:, ::-1][:, :min(dims, d)].copy()).to(device=x.device, dtype=torch.float32)  # leading components first
    proj = kernels.gemm(xc, axes.contiguous()) if on_gpu else xc @ axes   # [N, dims]
    lo, hi = proj.amin(0, keepdim=True), proj.amax(0, keepdim=True)
    q = ((proj - lo) / (hi - lo).clamp_min(1e-30) * (2**21 - 1)).to(torch.int64).clamp_(0, 2**21 - 1)

    def spread(v):  # 21 bits -> every third bit of 63
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        return (v | (v << 2)) & 0x1249249249249249
    code = torch.zeros(n, dtype=torch.int64, device=x.device)
    for a in range(q.shape[1]):
        code |= spread(q[:, a]) << (q.shape[1] - 1 - a)                 # the leading component owns the most significant bits
    return torch.sort(code, stable=True).indices


def locality_order(graph: CSRGraph, method: str = "rcm", coords: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A renumbering ``perm[new] = old`` under which neighbouring rows of a kNN-like graph gather mostly nearby rows, so that
    the SpMM's random 512-byte row reads hit L2 / the Infinity Cache instead of HBM ("cluster / kNN-BFS order";
    the same renumbering keeps halos small when the graph is sharded).  ``morton`` (needs ``coords``, the representation the graph's
    neighbours were searched in): Z-order over its leading principal components, on the device, milliseconds (``morton_order``).
    ``rcm``: reverse Cuthill-McKee of the symmetrised pattern (scipy on the host, O(nnz), 0.5 s at 1M cells) — for graphs that come
    without coordinates."""
    if method == "morton":
        if coords is None or coords.shape[0] != graph.n_rows:
            raise ValueError("locality_order(method='morton') needs coords [n_rows, d]")
        return morton_order(coords)
    if method != "rcm":
        raise ValueError(f"unknown locality order {method!r}")
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    import scipy.sparse as sp
    n = graph.n_rows
    a = sp.csr_matrix((np.ones(graph.nnz, dtype=np.int8), graph.col.cpu().numpy(), graph.rowptr.cpu().numpy()), shape=(n, graph.n_cols))
    if not graph.symmetric:
        a = (a + a.T).tocsr()
    return torch.from_numpy(np.ascontiguousarray(reverse_cuthill_mckee(a, symmetric_mode=True)).astype(np.int64))


# Keyed by IDENTITY (id + a finalizer that evicts the entry when the tensor dies).  A WeakKeyDictionary would compare
# keys with ``==`` on lookup, and ``aten::eq`` is not implemented for sparse tensors (the second call with the same
# adjacency raised).
_CACHE = {}


def _cache_get(adj):
    hit = _CACHE.get(id(adj))
    return hit[1] if hit is not None and hit[0]() is adj else None


def _cache_put(adj, g):
    key = id(adj)
    _CACHE[key] = (weakref.ref(adj), g)
    weakref.finalize(adj, _CACHE.pop, key, None)


class TensorKeyedCache:
    """One-entry-per-tensor cache of derived graph structures, keyed by tensor IDENTITY and ``_version``.

    A key built from ``data_ptr()`` + shape returns a stale structure after an in-place edit of the tensor and can alias a
    NEW tensor that the caching allocator placed in a freed block of the same shape.  Here an entry is hit
    only when the very same tensor object is passed again with an unchanged version counter; it is evicted when the tensor
    dies.  ``extra`` distinguishes derived structures that depend on more than the tensor (e.g. the node count)."""

    def __init__(self):
        self._d = {}

    def get(self, t: torch.Tensor, extra=None):
        hit = self._d.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == t._version and hit[2] == extra:
            return hit[3]
        return None

    def put(self, t: torch.Tensor, value, extra=None):
        key = id(t)
        if key not in self._d or self._d[key][0]() is not t:
            weakref.finalize(t, self._d.pop, key, None)
        self._d[key] = (weakref.ref(t), t._version, extra, value)
        return value

    def __len__(self):
        return len(self._d)


def as_graph(adj, device=None) -> CSRGraph:
    """Accept what the reference layers accept (torch sparse tensor) or a ready CSRGraph.

    Conversions are cached per adjacency object: the reference passes the same ``adj`` every epoch
    (scdsc.py:257-288), so the CSR and its transpose are built once.
    """
    if isinstance(adj, CSRGraph):
        return adj
    if isinstance(adj, torch.Tensor) and adj.layout in (torch.sparse_coo, torch.sparse_csr):
        g = _cache_get(adj)
        if g is None:
            g = CSRGraph.from_torch_sparse(adj, device)
            _cache_put(adj, g)
        return g
    raise TypeError(f"cannot interpret {type(adj)} as a sparse adjacency")
