"""Stand-in for the ``DGLGraph`` objects that cross the scDeepSort / graph-sc API.

DGL cannot be installed on ROCm here, and the reference's ``fit(graph, ...)`` / ``predict(graph)`` signatures take a
DGLGraph.  ``CellGeneGraph`` exposes the subset of that surface the hot-path call sites use (``ndata``/``edata``
frames, ``number_of_nodes``, ``in_degrees``, ``out_degrees``, ``edges``, ``in_edges``, ``subgraph``, ``to``) on top of
a device-resident CSR keyed by destination node; ``NeighborSampler`` / ``DataLoader`` reproduce the full-fan-out
in-neighbour blocks of ``dgl.dataloading.NeighborSampler([-1]*L)`` (scdeepsort.py:183,233-236; graphsc.py:181-183).

Edge order: ``eid`` maps every CSR slot to the edge's id in the graph's edge list (for the transform output this is
the reference's order, cell_feature_graph.py:43-69), so ``edges()`` / ``edata`` come back in that order.
"""
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


class _Frame(dict):
    """ndata / edata: name -> tensor."""


class _GatherFrame(dict):
    """srcdata / dstdata of a block: rows ``ids`` of the parent graph's node data, gathered on first access (a batch
    only ever touches ``features``, ``cell_id`` and ``label``; gathering every ndata entry per batch was a measurable
    part of the per-batch cost).  Entries assigned on the block live in the dict itself."""

    def __init__(self, parent: Dict[str, torch.Tensor], ids: torch.Tensor):
        super().__init__()
        self._parent, self._ids = parent, ids

    def __missing__(self, key):
        val = self._parent[key][self._ids]  # KeyError of the parent propagates
        self[key] = val
        return val

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._parent

    def keys(self):
        return list(dict.fromkeys(list(dict.keys(self)) + list(self._parent.keys())))

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())


class _EdgeView:
    """``edges.src`` / ``edges.dst`` of an edge batch: node-frame entries gathered per edge on access."""

    def __init__(self, frame, index: torch.Tensor):
        self._frame, self._index = frame, index

    def __getitem__(self, key):
        return self._frame[key][self._index]

    def __contains__(self, key):
        return key in self._frame


class EdgeBatch:
    """What a message UDF receives (``dgl.udf.EdgeBatch``): ``src[k]`` / ``dst[k]`` are the node features of every edge's end
    points, ``data[k]`` the edge features, all in the CSR slot order of the graph (destination-major)."""

    def __init__(self, src_frame, dst_frame, edata, src_index, dst_index):
        self.src, self.dst, self.data = _EdgeView(src_frame, src_index), _EdgeView(dst_frame, dst_index), edata
        self._n = int(src_index.numel())

    def batch_size(self) -> int:
        return self._n

    def __len__(self):
        return self._n


class _MessagePassing:
    """``update_all`` / ``adjacency_matrix`` for the graph and its blocks (gnn.py:90, graphsc.py:208,463-465): built-in
    ``fn.u_mul_e`` / ``fn.copy_u`` messages with ``fn.sum`` / ``fn.mean`` run as ONE dh_spmm_csr_f32 launch on the stored CSR; a Python
    message UDF (``AdaptiveSAGE.message_func``, ``edge_selection_simple``) gets an ``EdgeBatch`` and its [E, ...] message tensor is
    reduced per destination by the same kernel (the compatibility path: it materialises the messages, which the model code of this
    package never does).  Differentiable in the source features / the UDF's message (autograd.spmm), not in built-in edge weights.
    The frames read / written are ``srcdata`` / ``dstdata`` (= ``ndata`` on a whole graph)."""

    def _mp_parts(self):
        """(row pointers of the destination rows starting at 0, source index per slot, weights per slot, #dst, #src, frames)."""
        raise NotImplementedError

    def update_all(self, message_func, reduce_func, apply_node_func=None):
        from . import autograd, kernels
        from . import function as fn
        from .graph import CSRGraph
        if apply_node_func is not None:
            raise NotImplementedError("update_all: apply_node_func is not used by the reference's layers")
        if not isinstance(reduce_func, fn._Reduce):
            raise TypeError("update_all: the reduce function must be dance_amd.function.sum / mean")
        rowptr, col, val, n_dst, n_src, srcdata, dstdata, edata = self._mp_parts()
        reduce = kernels.REDUCE_MEAN if reduce_func.kind == "mean" else kernels.REDUCE_SUM
        if isinstance(message_func, fn._Message):
            if message_func.out != reduce_func.msg:
                raise KeyError(f"update_all: the reduce function reads {reduce_func.msg!r}, the message is {message_func.out!r}")
            z = srcdata[message_func.lhs]
            ew = None
            if message_func.kind == "u_mul_e":
                ew = edata[message_func.rhs]
                if ew.numel() != col.numel():
                    raise ValueError("update_all: u_mul_e needs one scalar per edge")
                ew = ew.reshape(-1).to(torch.float32).contiguous()
            shape = z.shape
            out = autograd.spmm(z.reshape(shape[0], -1).to(torch.float32), CSRGraph(rowptr, col, ew, n_dst, n_src), reduce=reduce)
            dstdata[reduce_func.out] = out.reshape((n_dst, ) + tuple(shape[1:]))
            return
        # message UDF: one EdgeBatch over all stored edges, then a segment reduction of its message tensor
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        dst_index = torch.repeat_interleave(torch.arange(n_dst, device=rowptr.device), deg)
        msgs = message_func(EdgeBatch(srcdata, dstdata, edata, col.to(torch.int64), dst_index))
        m = msgs[reduce_func.msg]
        e = int(col.numel())
        if m.shape[0] != e:
            raise ValueError(f"update_all: the message has {m.shape[0]} rows for {e} edges")
        slot = torch.arange(e, dtype=torch.int32, device=rowptr.device)
        out = autograd.spmm(m.reshape(e, -1).to(torch.float32), CSRGraph(rowptr, slot, None, n_dst, e), reduce=reduce)
        dstdata[reduce_func.out] = out.reshape((n_dst, ) + tuple(m.shape[1:]))

    def adjacency_matrix(self, transpose: bool = False):
        """Sparse [num_src, num_dst] matrix with a 1 per stored edge u -> v at (u, v) (``dgl.DGLGraph.adj`` of DGL 1.x: rows are
        sources; ``transpose=True`` puts the destinations in the rows); ``.to_dense()`` as graphsc.py:208 calls it."""
        rowptr, col, _, n_dst, n_src, *_ = self._mp_parts()
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        dst = torch.repeat_interleave(torch.arange(n_dst, device=rowptr.device), deg)
        src = col.to(torch.int64)
        idx, shape = (torch.stack((dst, src)), (n_dst, n_src)) if transpose else (torch.stack((src, dst)), (n_src, n_dst))
        return torch.sparse_coo_tensor(idx, torch.ones(src.numel(), dtype=torch.float32, device=rowptr.device), shape)

    adj = adjacency_matrix


class CellGeneGraph(_MessagePassing):

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, eid: Optional[torch.Tensor],
                 n_nodes: int, ndata: Optional[Dict[str, torch.Tensor]] = None):
        self.rowptr, self.col, self.val = rowptr, col, val  # int32, int32 (src ids), f32 — CSR by destination
        self.eid = eid  # int32 edge id of every CSR slot (None: slot order is the edge order)
        self._n_nodes = int(n_nodes)
        self.ndata = _Frame(ndata or {})
        self._edge_cache = None

    # ---- DGLGraph surface ----------------------------------------------------------------------------------
    def number_of_nodes(self) -> int:
        return self._n_nodes

    num_nodes = number_of_nodes

    def number_of_edges(self) -> int:
        return int(self.col.numel())

    num_edges = number_of_edges

    @property
    def device(self):
        return self.rowptr.device

    def nodes(self) -> torch.Tensor:
        return torch.arange(self._n_nodes, device=self.device)

    @property
    def srcdata(self):
        return self.ndata

    @property
    def dstdata(self):
        return self.ndata

    def _mp_parts(self):
        return self.rowptr, self.col, self.val, self._n_nodes, self._n_nodes, self.ndata, self.ndata, _Frame(weight=self.val[:, None])

    def add_edges(self, u, v, data=None):
        """Append the edges u[i] -> v[i] IN PLACE (``dgl.DGLGraph.add_edges``; cell_feature_graph.py:69 adds the self loops this way):
        they get the next edge ids, ``data["weight"]`` ([k] or [k, 1]; 0 when absent, as DGL zero-fills missing edge features)."""
        dev = self.device
        u = torch.as_tensor(u, dtype=torch.int64, device=dev).reshape(-1)
        v = torch.as_tensor(v, dtype=torch.int64, device=dev).reshape(-1)
        if u.numel() != v.numel():
            if u.numel() == 1:
                u = u.expand_as(v)
            elif v.numel() == 1:
                v = v.expand_as(u)
            else:
                raise ValueError("add_edges: u and v must have the same length (or one of them a single node)")
        if u.numel() and (int(torch.max(u.max(), v.max())) >= self._n_nodes or int(torch.min(u.min(), v.min())) < 0):
            raise ValueError("add_edges: node id out of range")
        extra = set((data or {}).keys()) - {"weight"}
        if extra:
            raise KeyError(f"add_edges: this graph stores one edge feature, 'weight' (got {sorted(extra)})")
        wt = (torch.as_tensor(data["weight"], dtype=torch.float32, device=dev).reshape(-1) if data and "weight" in data
              else torch.zeros(u.numel(), dtype=torch.float32, device=dev))
        if wt.numel() != u.numel():
            raise ValueError("add_edges: one weight per new edge")
        e_old = int(self.col.numel())
        eid_old = self.eid.to(torch.int64) if self.eid is not None else torch.arange(e_old, device=dev)
        src = torch.cat((self.col.to(torch.int64), u))
        dst = torch.cat((self._dst_of_slots(), v))
        eid = torch.cat((eid_old, torch.arange(e_old, e_old + u.numel(), device=dev)))
        val = torch.cat((self.val, wt))
        order = torch.argsort(dst * (e_old + u.numel() + 1) + eid)  # CSR by destination, a row's slots in edge-id order
        rowptr = torch.zeros(self._n_nodes + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=self._n_nodes), 0)
        self.rowptr, self.col, self.val = rowptr.to(torch.int32), src[order].to(torch.int32), val[order].contiguous()
        self.eid = eid[order].to(torch.int32)
        self._edge_cache = None
        for k in ("_zero_in_deg", "_gene_prefix", "_min_seed_cache"):
            self.__dict__.pop(k, None)

    def in_degrees(self) -> torch.Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)

    def out_degrees(self) -> torch.Tensor:
        return torch.bincount(self.col.to(torch.int64), minlength=self._n_nodes)

    def has_zero_in_degree(self) -> bool:
        if getattr(self, "_zero_in_deg", None) is None:
            self._zero_in_deg = bool((self.rowptr[1:] == self.rowptr[:-1]).any())
        return self._zero_in_deg

    def gene_prefix(self) -> int:
        """G if the nodes are laid out as CellFeatureGraph builds them — genes (cell_id >= 0) are nodes [0, G), every later
        node is a cell (cell_id == -1) — else -1.  One device read per graph, cached."""
        if getattr(self, "_gene_prefix", None) is None:
            if "cell_id" not in self.ndata:  # a graph without the CellFeatureGraph id columns: no known layout
                self._gene_prefix = -1
                return -1
            cid = self.ndata["cell_id"]
            g = int((cid >= 0).sum())
            self._gene_prefix = g if bool((cid[:g] >= 0).all()) else -1
        return self._gene_prefix

    def cell_rows_block(self) -> "Block":
        """The full-graph "block" whose destinations are ALL cells (rows [G, G+N) of the CSR, no copy, no sampling) and
        whose sources are all nodes: what the union of the sampler's cell batches computes, in one piece.  Needs the
        genes-first layout (``gene_prefix() >= 0``)."""
        g = self.gene_prefix()
        if g < 0:
            raise ValueError("cell_rows_block needs the CellFeatureGraph node layout (genes first, then cells)")
        n = self._n_nodes
        blk = Block.__new__(Block)
        blk.rowptr, blk.col, blk.val = self.rowptr[g:], self.col, self.val  # row pointers stay absolute offsets into col / val
        blk._num_src, blk._num_dst = n, n - g
        blk.parent, blk.dst_offset = self, g
        blk.gene_window = (0, g)  # every destination is a cell; their gene in-neighbours are source rows [0, g)
        ids = torch.arange(n, device=self.device)
        blk.srcdata = _Frame(self.ndata)
        blk.srcdata["_ID"] = ids
        blk.dstdata = _Frame({k: v[g:] for k, v in self.ndata.items()})
        blk.dstdata["_ID"] = ids[g:]
        blk.edata = _Frame()
        return blk

    def all_rows_block(self) -> "Block":
        """The full-graph "block" whose destinations are ALL nodes (the CSR as it is): an inner layer of a multi-layer pass over the
        whole graph — what the sampler's fanout -1 blocks (scdeepsort.py:183 ``[-1] * n_layers``) add up to.  With the genes-first
        layout the gene rows [0, G) take their in-edges from the cell window [G, G + N) (``cell_window``, the split-K matrix-core
        kernel) and the cell rows from the gene window [0, G) (``gene_window``); ``gene_rows`` = G."""
        g = self.gene_prefix()
        if g < 0:
            raise ValueError("all_rows_block needs the CellFeatureGraph node layout (genes first, then cells)")
        n = self._n_nodes
        blk = Block.__new__(Block)
        blk.rowptr, blk.col, blk.val = self.rowptr, self.col, self.val
        blk._num_src, blk._num_dst = n, n
        blk.parent, blk.dst_offset = self, 0
        blk.gene_window, blk.cell_window, blk.gene_rows = (0, g), (g, n - g), g
        ids = torch.arange(n, device=self.device)
        blk.srcdata = _Frame(self.ndata)
        blk.srcdata["_ID"] = ids
        blk.dstdata = _Frame(self.ndata)
        blk.dstdata["_ID"] = ids
        blk.edata = _Frame()
        return blk

    def _dst_of_slots(self) -> torch.Tensor:
        deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        return torch.repeat_interleave(torch.arange(self._n_nodes, device=self.device), deg)

    def _edge_lists(self):
        """(src, dst, weight) in edge-id order."""
        if self._edge_cache is None:
            src, dst, w = self.col.to(torch.int64), self._dst_of_slots(), self.val
            if self.eid is not None:
                order = torch.argsort(self.eid.to(torch.int64))
                src, dst, w = src[order], dst[order], w[order]
            self._edge_cache = (src, dst, w)
        return self._edge_cache

    def edges(self):
        src, dst, _ = self._edge_lists()
        return src, dst

    @property
    def edata(self) -> Dict[str, torch.Tensor]:
        return _Frame(weight=self._edge_lists()[2][:, None])

    def in_edges(self, v: int, form: str = "uv"):
        s, t = int(self.rowptr[v]), int(self.rowptr[v + 1])
        src = self.col[s:t].to(torch.int64)
        dst = torch.full_like(src, v)
        ids = (self.eid[s:t] if self.eid is not None else torch.arange(s, t, device=self.device)).to(torch.int64)
        return (src, dst, ids) if form == "all" else (src, dst)

    def to(self, device) -> "CellGeneGraph":
        if torch.device(device) == self.device:
            return self
        g = CellGeneGraph(self.rowptr.to(device), self.col.to(device), self.val.to(device),
                          None if self.eid is None else self.eid.to(device), self._n_nodes,
                          {k: v.to(device) for k, v in self.ndata.items()})
        return g

    def with_ndata(self, **updates) -> "CellGeneGraph":
        """Shallow copy (same structure tensors) with some node-data entries replaced."""
        return CellGeneGraph(self.rowptr, self.col, self.val, self.eid, self._n_nodes, {**self.ndata, **updates})

    def local_scope(self):
        import contextlib
        return contextlib.nullcontext()

    def subgraph(self, nodes) -> "CellGeneGraph":
        """Induced subgraph; node i of the result is ``nodes[i]`` and edges keep their relative order
        (``dgl.DGLGraph.subgraph``, examples/single_modality/cell_type_annotation/scdeepsort.py:65-66)."""
        dev = self.device
        nodes = torch.as_tensor(nodes, dtype=torch.int64, device=dev)
        lut = torch.full((self._n_nodes, ), -1, dtype=torch.int64, device=dev)
        lut[nodes] = torch.arange(nodes.numel(), device=dev)
        dst = self._dst_of_slots()
        src = self.col.to(torch.int64)
        keep = (lut[src] >= 0) & (lut[dst] >= 0)
        new_src, new_dst = lut[src[keep]], lut[dst[keep]]
        old_eid = (self.eid.to(torch.int64) if self.eid is not None else torch.arange(src.numel(), device=dev))[keep]
        val = self.val[keep]
        # CSR by new destination, slots inside a row ordered by original edge id; new edge ids = rank of old ids
        order = torch.argsort(new_dst * (int(old_eid.max()) + 1 if old_eid.numel() else 1) + old_eid)
        new_src, new_dst, val, old_eid = new_src[order], new_dst[order], val[order], old_eid[order]
        rank = torch.empty_like(old_eid)
        rank[torch.argsort(old_eid)] = torch.arange(old_eid.numel(), device=dev)
        rowptr = torch.zeros(nodes.numel() + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(new_dst, minlength=nodes.numel()), 0)
        g = CellGeneGraph(rowptr.to(torch.int32), new_src.to(torch.int32), val.contiguous(), rank.to(torch.int32),
                          nodes.numel(), {k: v[nodes] for k, v in self.ndata.items()})
        g.ndata["_ID"] = nodes
        return g


class Block(_MessagePassing):
    """Bipartite message-flow block: ``num_dst`` destination nodes (= the first ``num_dst`` source nodes) with all
    their in-edges; CSR rows are destinations, columns index ``srcdata`` rows."""

    def __init__(self, rowptr, col, val, num_src: int, num_dst: int, src_ids: torch.Tensor, parent: CellGeneGraph):
        self.rowptr, self.col, self.val = rowptr, col, val
        self._num_src, self._num_dst = int(num_src), int(num_dst)
        self.parent = parent
        self.dst_offset = 0  # destination node i is source node dst_offset + i (0 for sampled blocks: dgl.to_block order)
        self.gene_window = None  # (first source row, count) of the gene rows when every destination is known to be a cell
        self.srcdata = _GatherFrame(parent.ndata, src_ids)
        self.srcdata["_ID"] = src_ids
        self.dstdata = _GatherFrame(parent.ndata, src_ids[:num_dst])  # destination nodes = the first num_dst sources
        self.dstdata["_ID"] = src_ids[:num_dst]
        self.edata = _Frame(weight=val[:, None])

    def number_of_dst_nodes(self) -> int:
        return self._num_dst

    num_dst_nodes = number_of_dst_nodes

    def _mp_parts(self):
        rp = self.rowptr_dst
        first = int(rp[0]) if rp.numel() else 0  # the full-graph cell block keeps absolute offsets into the graph's col / val
        if first:
            last = int(rp[-1])
            return rp - first, self.col[first:last], self.val[first:last], self._num_dst, self._num_src, self.srcdata, self.dstdata, \
                _Frame(weight=self.val[first:last, None])
        e = int(rp[-1]) if rp.numel() else 0
        return rp, self.col[:e], self.val[:e], self._num_dst, self._num_src, self.srcdata, self.dstdata, _Frame(weight=self.val[:e, None])

    @property
    def rowptr_dst(self) -> torch.Tensor:
        """Row pointers of exactly the destination rows (``num_dst + 1`` entries; a StaticCellBlock's CSR has one padding row more)."""
        return self.rowptr[:self._num_dst + 1]

    def number_of_src_nodes(self) -> int:
        return self._num_src

    num_src_nodes = number_of_src_nodes

    def number_of_edges(self) -> int:
        return int(self.col.numel())

    def in_degrees(self) -> torch.Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)

    def out_degrees(self) -> torch.Tensor:
        # scatter-add instead of torch.bincount: bincount reads max(col) back to the host to size its output
        out = torch.zeros(self._num_src, dtype=torch.int64, device=self.col.device)
        return out.index_add_(0, self.col.to(torch.int64), torch.ones(1, dtype=torch.int64, device=self.col.device).expand(self.col.numel()))

    def has_zero_in_degree(self) -> bool:
        """Whether some destination node has no in-edge.  Answered from the parent graph when that has none at all (one
        device read per GRAPH, cached) — only otherwise from this block's own rows (a device read per block)."""
        if not self.parent.has_zero_in_degree():
            return False
        return bool((self.rowptr[1:] == self.rowptr[:-1]).any())

    def to(self, device) -> "Block":
        return self  # blocks are created on the graph's device

    def local_scope(self):
        import contextlib
        return contextlib.nullcontext()

    def dstnodes(self):
        return torch.arange(self._num_dst, device=self.rowptr.device)


class StaticCellBlock(Block):
    """The block of ``batch`` seed CELLS of a CellFeatureGraph-layout graph with STATIC shapes (dh_block_cells_static): sources =
    [the seeds | all genes], exactly ``e_max`` stored entries, one padding destination row (index ``batch``) over the unused tail.
    The buffers are owned by the block and refilled by ``rebuild()`` — a handful of launches without any host round trip, so a
    whole training step over it can be captured as ONE hipGraph (``GraphSC.fit`` / ``ScDeepSort.fit``).  Layer outputs equal those
    of the dgl.to_block-ordered block: the extra source rows (genes no seed expresses) have no edge."""

    def __init__(self, parent: CellGeneGraph, batch: int):
        from . import kernels
        g = parent.gene_prefix()
        if g < 0:
            raise ValueError("StaticCellBlock needs the CellFeatureGraph node layout (genes first, then cells)")
        dev = parent.device
        deg = (parent.rowptr[g + 1:] - parent.rowptr[g:-1])
        self.n_genes, self.batch = g, int(batch)
        self.e_max = int(batch) * int(deg.max()) if deg.numel() else 0   # one host read per (graph, batch size), at set-up
        self.seeds = torch.zeros(batch, dtype=torch.int64, device=dev)
        self.seeds.copy_(torch.arange(g, g + batch, device=dev) if parent.number_of_nodes() >= g + batch else torch.full((batch, ), g, device=dev))
        self.src_ids = torch.cat((self.seeds, torch.arange(g, dtype=torch.int64, device=dev)))  # tail constant, head = seeds (rebuild)
        self.rowptr = torch.zeros(batch + 2, dtype=torch.int32, device=dev)
        self.col = torch.zeros(self.e_max, dtype=torch.int32, device=dev)
        self.val = torch.zeros(self.e_max, dtype=torch.float32, device=dev)
        self.bad = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ws = torch.empty(max(kernels.block_cells_static_workspace_bytes(batch), 1), dtype=torch.uint8, device=dev)
        self._num_src, self._num_dst = batch + g, int(batch)
        self.parent, self.dst_offset = parent, 0
        self.gene_window = (batch, g)
        self.pad_row = True  # CSR rows = num_dst + 1; consumers slice their output to num_dst rows
        self.srcdata = _GatherFrame(parent.ndata, self.src_ids)
        self.srcdata["_ID"] = self.src_ids
        self.dstdata = _GatherFrame(parent.ndata, self.seeds)
        self.dstdata["_ID"] = self.seeds
        self.edata = _Frame(weight=self.val[:, None])

    def rebuild(self):
        """Refill the buffers for the ids currently in ``self.seeds`` (device only; capturable)."""
        from . import kernels
        from .capture import kernel_copy_
        kernel_copy_(self.src_ids[:self.batch], self.seeds)  # a kernel, not a memcpy node (capture.py)
        kernels.block_cells_static(self.parent.rowptr, self.parent.col, self.parent.val, self.seeds, self.n_genes, self.rowptr, self.col, self.val,
                                   self.bad, self._ws)
        self.__dict__.pop("_wgc_scales", None)  # per-batch caches of the layers
        self.__dict__.pop("_csr", None)
        for frame in (self.srcdata, self.dstdata):  # gathered node data of the previous batch
            for k in [k for k in dict.keys(frame) if k != "_ID"]:
                dict.__delitem__(frame, k)
        return self

    def number_of_edges(self) -> int:
        return self.e_max

    def in_degrees(self) -> torch.Tensor:
        return (self.rowptr[1:self.batch + 1] - self.rowptr[:self.batch]).to(torch.int64)

    def out_degrees(self) -> torch.Tensor:
        valid = (torch.arange(self.e_max, device=self.col.device) < self.rowptr[self.batch]).to(torch.int64)  # the padding tail counts nothing
        return torch.zeros(self._num_src, dtype=torch.int64, device=self.col.device).index_add_(0, self.col.to(torch.int64), valid)

    def has_zero_in_degree(self) -> bool:
        return False  # every cell has its self loop


def _full_in_block(g: CellGeneGraph, seeds: torch.Tensor) -> Block:
    """All in-edges of ``seeds``; source nodes = seeds first, then the remaining in-neighbours (ascending id).  Built by
    dh_block_plan / dh_block_fill (block.hip); the node bitmap and the node -> position table live on the graph and are
    reused by every batch (the bitmap is handed back all-zero by the kernels, nothing is memset per batch)."""
    from . import kernels
    dev = g.device
    seeds = seeds.to(dev).to(torch.int64).contiguous()
    scratch = getattr(g, "_block_scratch", None)
    if scratch is None:
        scratch = (torch.zeros(g.number_of_nodes(), dtype=torch.uint8, device=dev),
                   torch.empty(g.number_of_nodes(), dtype=torch.int32, device=dev))
        g._block_scratch = scratch
    brp, bcol, bval, src_ids = kernels.block_build(g.rowptr, g.col, g.val, seeds, *scratch)
    return Block(brp, bcol, bval, src_ids.numel(), seeds.numel(), src_ids, g)


def _sampled_in_block(g: CellGeneGraph, seeds: torch.Tensor, fanout: int, generator=None) -> Block:
    """At most ``fanout`` in-edges per seed, drawn uniformly WITHOUT replacement (``dgl.sampling.sample_neighbors(g, seeds, fanout,
    edge_dir="in")``; a seed with fewer in-edges keeps them all), as a message-flow block with dgl.to_block's numbering.  A random
    key per in-edge of the seeds, a sort by (seed, key), the first ``fanout`` of every seed kept, the kept edges put back into the
    graph's own edge order; the block itself comes from the same builder as the full-neighbour blocks, fed a CSR that holds the kept
    edges only.  (The draws are torch's, not DGL's: which edges are kept is not comparable with a DGL run, the distribution is.)"""
    from . import kernels
    dev = g.device
    seeds = seeds.to(dev).to(torch.int64).contiguous()
    n = g.number_of_nodes()
    rp = g.rowptr.to(torch.int64)
    start, deg = rp[seeds], rp[seeds + 1] - rp[seeds]
    seg = torch.repeat_interleave(torch.arange(seeds.numel(), device=dev), deg)
    first = torch.cumsum(deg, 0) - deg                                   # offset of every seed's run in the concatenated edge list
    e = start[seg] + (torch.arange(seg.numel(), device=dev) - first[seg])  # positions in g.col / g.val
    key = seg.to(torch.float64) + torch.rand(seg.numel(), device=dev, generator=generator, dtype=torch.float64)
    order = torch.argsort(key)
    rank = torch.arange(seg.numel(), device=dev) - first[seg[order]]
    kept = order[rank < fanout]
    # a CSR over ALL nodes with the kept edges in the rows of their seeds (ascending node id, edge order inside a row as in the graph)
    node = seeds[seg[kept]]
    srt = torch.argsort(node * g.col.numel() + e[kept])
    ek = e[kept][srt]
    counts = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, node, torch.ones_like(node))
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    scratch = getattr(g, "_block_scratch", None)
    if scratch is None:
        scratch = (torch.zeros(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.int32, device=dev))
        g._block_scratch = scratch
    brp, bcol, bval, src_ids = kernels.block_build(rowptr, g.col[ek].contiguous(), g.val[ek].contiguous(), seeds, *scratch)
    blk = Block(brp, bcol, bval, src_ids.numel(), seeds.numel(), src_ids, g)
    # the kept edges in block order (block rows follow the seeds' order, a row's edges the graph's): their ids in the parent graph
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    pos[seeds] = torch.arange(seeds.numel(), device=dev)
    slots = ek[torch.argsort(pos[node[srt]], stable=True)]
    # "_ID" = the parent graph's EDGE ids, as DGL's blocks carry them (index edata / edge-id-ordered arrays with it); "_SLOT" = the
    # positions in the parent's CSR arrays (g.col / g.val).  The two coincide only on a graph whose CSR slots are in edge order
    # (g.eid is None); a CellFeatureGraph's are not (its edge order is the reference's: cell->gene, gene->cell, self loops).
    blk.edata["_SLOT"] = slots
    blk.edata["_ID"] = slots if g.eid is None else g.eid[slots].to(torch.int64)
    return blk


class NeighborSampler:
    """``dgl.dataloading.NeighborSampler(fanouts, edge_dir="in")``: one block per entry of ``fanouts``, the LAST entry for the hop next
    to the seeds (DGL walks ``reversed(fanouts)``).  -1 = every in-neighbour — all the reference uses (scdeepsort.py:183,
    graphsc.py:181 MultiLayerFullNeighborSampler) and the only form the matrix-core aggregation and the captured steps take; a positive
    entry draws that many in-edges per node uniformly without replacement (``generator``: a torch.Generator on the graph's device for
    reproducible draws)."""

    def __init__(self, fanouts: Sequence[int], edge_dir: str = "in", prob=None, mask=None, replace: bool = False, generator=None, **_ignored):
        if edge_dir != "in":
            raise NotImplementedError("edge_dir='out' is not used by the hot path (every caller aggregates over in-edges)")
        if prob is not None or mask is not None or replace:
            raise NotImplementedError("weighted / masked / with-replacement neighbour sampling is not used by the hot path")
        if any(int(f) == 0 or int(f) < -1 for f in fanouts):
            raise ValueError(f"fanouts must be -1 (all in-neighbours) or positive, got {list(fanouts)}")
        self.fanouts = [int(f) for f in fanouts]
        self.num_layers = len(self.fanouts)
        self.generator = generator

    def sample(self, g: CellGeneGraph, seeds: torch.Tensor, cells_only: bool = False):
        blocks: List[Block] = []
        out_nodes = seeds
        for layer in range(self.num_layers):
            fan = self.fanouts[self.num_layers - 1 - layer]
            blk = _full_in_block(g, seeds) if fan == -1 else _sampled_in_block(g, seeds, fan, self.generator)
            if cells_only and layer == 0:  # destinations = the seed cells, remaining sources = their genes (ascending id)
                blk.gene_window = (blk.number_of_dst_nodes(), blk.number_of_src_nodes() - blk.number_of_dst_nodes())
            blocks.insert(0, blk)
            seeds = blk.srcdata["_ID"]
        return seeds, out_nodes, blocks


def MultiLayerFullNeighborSampler(num_layers: int):
    return NeighborSampler([-1] * num_layers)


class DataLoader:
    """Mini-batch iterator yielding ``(input_nodes, output_nodes, blocks)`` like ``dgl.dataloading.DataLoader``."""

    def __init__(self, graph: CellGeneGraph, indices, sampler: Optional[NeighborSampler] = None, batch_size: int = 1,
                 shuffle: bool = False, drop_last: bool = False, generator: Optional[torch.Generator] = None, prefetch: bool = True,
                 block_hook=None, graph_sampler: Optional[NeighborSampler] = None, **_ignored):
        # ``graph_sampler`` is dgl.dataloading.DataLoader's own name for the third argument (scdeepsort.py:233,266,318 pass it by
        # keyword); ``sampler`` is kept for this package's earlier call sites
        sampler = graph_sampler if sampler is None else sampler
        if sampler is None:
            raise TypeError("DataLoader: a graph_sampler is required")
        self.graph, self.sampler = graph, sampler
        # block_hook(blocks) runs right after a batch's blocks are built, on the stream that built them (the side stream when
        # prefetching): per-batch set-up whose results the host needs (e.g. a count) then never waits for the model's kernels.
        # Its return value is stored as blocks[-1].hook_out.
        self.block_hook = block_hook
        self.indices = torch.as_tensor(indices, dtype=torch.int64).to(graph.device)  # seeds live where the graph lives
        self.batch_size, self.shuffle, self.drop_last, self.generator = batch_size, shuffle, drop_last, generator
        self.prefetch = prefetch
        # seeds that are all cells of a CellFeatureGraph-layout graph: a one-layer block then is [seed cells | their genes,
        # ascending], which lets AdaptiveSAGE aggregate on the matrix cores (one device read per loader, not per batch)
        g = graph.gene_prefix() if hasattr(graph, "gene_prefix") else -1
        self.cells_only = bool(g >= 0 and self.indices.numel() > 0 and self._min_seed(graph, indices) >= g)

    def _min_seed(self, graph, given) -> int:
        """min(indices).  Host seeds (lists, numpy, CPU tensors) are reduced on the host — no device read at all.  A device tensor
        is read once per (graph, tensor OBJECT, version): ScDeepSort builds a new loader every epoch over the same ids.  The key is
        the caller's tensor identity + ``_version`` (graph.TensorKeyedCache), never ``data_ptr()``: ``self.indices`` is a fresh
        copy per loader and the caching allocator hands a freed block to the next allocation of the same size, so a pointer key
        could return another seed set's minimum."""
        if not (torch.is_tensor(given) and given.is_cuda):
            host = given.detach().cpu().numpy() if torch.is_tensor(given) else np.asarray(given)
            return int(host.min())
        from .graph import TensorKeyedCache
        cache = graph.__dict__.setdefault("_min_seed_cache", None)
        if not isinstance(cache, TensorKeyedCache):
            cache = graph.__dict__["_min_seed_cache"] = TensorKeyedCache()
        hit = cache.get(given)
        if hit is None:
            hit = cache.put(given, int(given.min()))
        return hit

    def enable_cpu_affinity(self, *args, **kwargs):
        """dgl.dataloading.DataLoader.enable_cpu_affinity (scdeepsort.py:236 wraps a CPU epoch in it): pins DGL's worker
        processes to cores.  There are no worker processes here (blocks are built by device kernels), so: a no-op context."""
        import contextlib
        return contextlib.nullcontext()

    def __len__(self):
        n = self.indices.numel()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        idx = self.indices
        if self.shuffle:
            # shuffled on the device (a host randperm of 1M seeds cost more than a whole epoch of kernels); a user-supplied
            # generator keeps its own (host) stream for reproducibility
            perm = (torch.randperm(idx.numel(), device=idx.device) if self.generator is None else
                    torch.randperm(idx.numel(), generator=self.generator).to(idx.device))
            idx = idx[perm]
        n = len(self)
        if not (self.prefetch and idx.is_cuda and n > 1):
            for i in range(n):
                out = self.sampler.sample(self.graph, idx[i * self.batch_size:(i + 1) * self.batch_size], self.cells_only)
                if self.block_hook is not None:
                    out[2][-1].hook_out = self.block_hook(out[2])
                yield out
            return
        # Blocks are built one batch ahead on a side stream: the builder's only host round trip (the size read between
        # dh_block_plan and dh_block_fill) then waits for a handful of small kernels instead of for the model's forward /
        # backward of the previous batch still running on the caller's stream — the host keeps enqueuing, the GPU never idles.
        main = torch.cuda.current_stream(idx.device)
        side = _side_stream(idx.device)
        side.wait_stream(main)  # the (shuffled) seeds and the graph are ready

        def build(i):
            with torch.cuda.stream(side):
                out = self.sampler.sample(self.graph, idx[i * self.batch_size:(i + 1) * self.batch_size], self.cells_only)
                if self.block_hook is not None:
                    out[2][-1].hook_out = self.block_hook(out[2])
                ev = torch.cuda.Event()
                ev.record(side)
            return out, ev

        try:
            nxt = build(0)
            for i in range(n):
                (inp, outn, blocks), ev = nxt
                if i + 1 < n:
                    nxt = build(i + 1)
                main.wait_event(ev)
                for blk in blocks:  # allocated on the side stream, consumed on the caller's
                    hook_out = getattr(blk, "hook_out", None)
                    extra = tuple(t for t in hook_out if torch.is_tensor(t) and t.is_cuda) if isinstance(hook_out, (tuple, list)) else ()
                    for t in (blk.rowptr, blk.col, blk.val, blk.srcdata["_ID"]) + extra:
                        if t is not None:
                            t.record_stream(main)
                yield inp, outn, blocks
        finally:
            # a consumer that stops early (break, exception) leaves the batch built ahead pending on the side stream, and the
            # builder's per-graph scratch (mark / lut, "all zero between calls") possibly in use: order the caller's stream after
            # it, so that a later build on the main stream (prefetch=False, a one-batch loader) neither races nor sees it dirty
            main.wait_stream(side)


_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]
