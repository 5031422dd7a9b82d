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


class CellGeneGraph:

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


class Block:
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
        self.src_ids[:self.batch].copy_(self.seeds)
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


class NeighborSampler:
    """``dgl.dataloading.NeighborSampler(fanouts, edge_dir="in")`` restricted to full fan-out (every entry -1),
    which is all the reference uses (scdeepsort.py:183, graphsc.py:181 MultiLayerFullNeighborSampler)."""

    def __init__(self, fanouts: Sequence[int], edge_dir: str = "in"):
        if any(f != -1 for f in fanouts) or edge_dir != "in":
            raise NotImplementedError("only full in-neighbour sampling ([-1]*L, edge_dir='in') is used by the hot path")
        self.num_layers = len(fanouts)

    def sample(self, g: CellGeneGraph, seeds: torch.Tensor, cells_only: bool = False):
        blocks: List[Block] = []
        out_nodes = seeds
        for layer in range(self.num_layers):
            blk = _full_in_block(g, seeds)
            if cells_only and layer == 0:  # destinations = the seed cells, remaining sources = their genes (ascending id)
                blk.gene_window = (blk.number_of_dst_nodes(), blk.number_of_src_nodes() - blk.number_of_dst_nodes())
            blocks.insert(0, blk)
            seeds = blk.srcdata["_ID"]
        return seeds, out_nodes, blocks


def MultiLayerFullNeighborSampler(num_layers: int):
    return NeighborSampler([-1] * num_layers)


class DataLoader:
    """Mini-batch iterator yielding ``(input_nodes, output_nodes, blocks)`` like ``dgl.dataloading.DataLoader``."""

    def __init__(self, graph: CellGeneGraph, indices, sampler: NeighborSampler, batch_size: int = 1,
                 shuffle: bool = False, drop_last: bool = False, generator: Optional[torch.Generator] = None, prefetch: bool = True,
                 block_hook=None, **_ignored):
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
        self.cells_only = bool(g >= 0 and self.indices.numel() > 0 and self._min_seed(graph) >= g)

    def _min_seed(self, graph) -> int:
        """min(indices), read from the device once per (graph, seed tensor): ScDeepSort builds a new loader every epoch over
        the same ids."""
        cache = graph.__dict__.setdefault("_min_seed_cache", {})
        key = (self.indices.data_ptr(), self.indices.numel())
        if key not in cache:
            cache.clear()
            cache[key] = int(self.indices.min())
        return cache[key]

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
