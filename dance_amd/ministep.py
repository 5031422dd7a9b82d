"""Host side of the persistent mini-batch steps (csrc/ministep.hip): ``GraphSC.fit`` / ``ScDeepSort.fit`` at the reference's default batch
sizes (graphsc.py:181-230, batch 128; scdeepsort.py:222-257, batch 500) as runs of steps behind ONE C call each — four kernel launches
per step, no framework op, no host read.

``GraphSCStepper`` / ``ScDeepSortStepper`` bind a model + its ``torch.optim.Adam`` + a CellFeatureGraph-layout graph to the C structs
(``dh_graphsc_step_t`` / ``dh_scdeepsort_step_t``): the parameters, their Adam moments and step counters are the optimiser's OWN tensors,
updated in place, so an eager step in between (the short last batch of an epoch), ``state_dict()`` and checkpoints see what torch would
have written.  More than one process: every step is gradients (phase 1) -> one flat all-reduce -> update (phase 2).
"""
import ctypes
from ctypes import c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p
from typing import Optional

import torch

from . import _lib


class AdamState(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("step", c_void_p)]


class GraphSCStep(ctypes.Structure):
    """``dh_graphsc_step_t`` (include/dance_hip.h)."""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("features", c_void_p),
                ("ld_features", c_int64), ("n_nodes", c_int64), ("n_genes", c_int64),
                ("batch", c_int64), ("in_feats", c_int64), ("hidden", c_int64), ("emb", c_int64),
                ("agg_mean", c_int32), ("phase", c_int32), ("max_row_entries", c_int32), ("reserved", c_int32),
                ("w1", AdamState), ("b1", AdamState), ("w2", AdamState), ("b2", AdamState),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
                ("dropout", c_float), ("decoder_dropout", c_float),
                ("seed", c_uint64), ("step0", c_uint64),
                ("seeds", c_void_p), ("z_out", c_void_p), ("loss_out", c_void_p), ("bad", c_void_p), ("grads", c_void_p), ("ax_out", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class ScDeepSortStep(ctypes.Structure):
    """``dh_scdeepsort_step_t`` (include/dance_hip.h)."""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("features", c_void_p),
                ("ld_features", c_int64), ("n_nodes", c_int64), ("n_genes", c_int64),
                ("cell_id", c_void_p), ("labels", c_void_p), ("alpha", c_void_p),
                ("batch", c_int64), ("dim_in", c_int64), ("hidden", c_int64), ("n_classes", c_int64),
                ("features_bf16", c_int32), ("phase", c_int32),
                ("w1", AdamState), ("b1", AdamState), ("w2", AdamState), ("b2", AdamState),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("weight_decay", c_float),
                ("dropout", c_float),
                ("seed", c_uint64), ("step0", c_uint64),
                ("seeds", c_void_p), ("loss_out", c_void_p), ("neigh_out", c_void_p), ("bad", c_void_p), ("grads", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


def adam_plain(optim) -> bool:
    """The optimiser configurations the fused update reproduces: one parameter group of ``torch.optim.Adam`` with float hyper-parameters,
    no amsgrad / maximize."""
    if type(optim) is not torch.optim.Adam or len(optim.param_groups) != 1:
        return False
    g = optim.param_groups[0]
    return not (g.get("amsgrad") or g.get("maximize") or g.get("differentiable")) and isinstance(g["lr"], (int, float))


def ensure_adam_state(optim, params):
    """Create the state ``torch.optim.Adam`` would create on its first step (zero moments, a float32 device step counter: what the
    fused / capturable implementations keep) for parameters that have none yet."""
    for p in params:
        st = optim.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not (torch.is_tensor(st.get("step")) and st["step"].is_cuda and st["step"].dtype == torch.float32):
            raise _lib.DanceHipError("the optimiser keeps host-side step counters: build it with fused=True (or capturable=True)")


def _adam_struct(optim, p) -> AdamState:
    st = optim.state[p]
    for t in (p, st["exp_avg"], st["exp_avg_sq"]):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise _lib.DanceHipError("the persistent step needs contiguous fp32 parameters and Adam moments on the GPU")
    return AdamState(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr())


def _draw_seed() -> int:
    """The dropout key of a fit: one draw from torch's default generator (``torch.manual_seed`` reproduces the masks)."""
    return int(torch.randint(0, 2**62, (1, ), dtype=torch.int64))


class _Stepper:
    """Common part: the hyper-parameters, the step counter of the dropout key, the flags, the data-parallel phases."""

    def _hyper(self, cfg, optim):
        g = optim.param_groups[0]
        cfg.lr, (cfg.beta1, cfg.beta2), cfg.eps, cfg.weight_decay = float(g["lr"]), map(float, g["betas"]), float(g["eps"]), float(g["weight_decay"])

    def check_flags(self, what: str, mask: int = 7):
        flags = int(self.bad) & mask
        if flags:
            self.bad.zero_()
            why = []
            if flags & 1:
                why.append("a seed is not a cell row of a CellFeatureGraph-layout graph (genes first; a cell's in-neighbours = its genes + its own self loop)")
            if flags & 2:
                why.append("a seed without exactly one self loop (the decoder target among a batch's own cells is taken to be the identity)")
            if flags & 4:
                why.append("a label outside [0, n_classes)")
            raise RuntimeError(f"{what}: " + "; ".join(why) + " — set DANCE_AMD_MINISTEP=0 for the general loop")

    def _run(self, fn, tag, first, n):
        from . import kernels
        kernels._call(tag, fn, ctypes.byref(self.cfg), int(first), int(n), kernels._stream())

    def run(self, seeds: torch.Tensor, n_steps: int, *outs):
        """``n_steps`` consecutive steps over ``seeds`` (int64 [>= n_steps * batch], on the device)."""
        raise NotImplementedError


class GraphSCStepper(_Stepper):
    """``GraphSC.fit``'s batch loop (graphsc.py:196-219) on dh_graphsc_steps."""

    @staticmethod
    def eligible(model, g, batch_size: int, optim) -> bool:
        import torch.nn as nn
        import torch.nn.functional as F

        from .autograd import HipLinear
        if not adam_plain(optim) or not hasattr(g, "gene_prefix") or g.gene_prefix() < 1 or g.device.type != "cuda":
            return False
        if hasattr(model, "layer2") or model.hidden is None or len(model.hidden) != 1 or len(model.encoder) != 1 or type(model.encoder[0]) is not HipLinear:
            return False
        l1 = model.layer1
        relu = l1._activation in (F.relu, torch.relu) or isinstance(l1._activation, nn.ReLU)
        if not relu or l1._norm != "both" or l1.weight is None or l1.bias is None or model.encoder[0].bias is None or model.agg not in ("sum", "mean"):
            return False
        if not getattr(model.decoder, "linear_logits", False):
            return False
        feats = g.ndata["features"]
        if feats.dtype != torch.float32 or feats.dim() != 2 or feats.stride(1) != 1:
            return False
        return bool(_lib.load().dh_graphsc_step_supported(int(batch_size), l1._in_feats, l1._out_feats, model.encoder[0].out_features))

    def __init__(self, model, g, batch_size: int, optim, world: int = 1):
        lib = _lib.load()
        _lib.require_device()
        self.model, self.optim, self.g, self.batch, self.world = model, optim, g, int(batch_size), int(world)
        l1, enc = model.layer1, model.encoder[0]
        self.params = [l1.weight, l1.bias, enc.weight, enc.bias]
        ensure_adam_state(optim, self.params)
        feats = g.ndata["features"]
        ng, f, h, e = g.gene_prefix(), l1._in_feats, l1._out_feats, enc.out_features
        dev = feats.device
        self.emb_dim = e
        self.bad = torch.zeros(1, dtype=torch.int32, device=dev)
        nbytes = int(lib.dh_graphsc_step_workspace_bytes(ng, self.batch, f, h, e))
        self.ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        self.grads = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev) if world > 1 else None
        c = self.cfg = GraphSCStep()
        c.rowptr, c.col, c.val, c.features = g.rowptr.data_ptr(), g.col.data_ptr(), g.val.data_ptr(), feats.data_ptr()
        c.ld_features, c.n_nodes, c.n_genes = feats.stride(0), g.number_of_nodes(), ng
        c.batch, c.in_feats, c.hidden, c.emb = self.batch, f, h, e
        c.agg_mean = 1 if model.agg == "mean" else 0
        # the longest cell row (one host read per stepper): short rows let the large-batch aggregation run as a dense product (phase 3)
        c.max_row_entries = int((g.rowptr[ng + 1:] - g.rowptr[ng:-1]).max()) if self.batch >= 1024 and g.number_of_nodes() > ng else 0
        c.w1, c.b1, c.w2, c.b2 = (_adam_struct(optim, p) for p in self.params)
        self._hyper(c, optim)
        c.dropout = float(model.dropout.p) if model.dropout is not None and model.training else 0.0
        c.decoder_dropout = float(model.decoder.dropout)  # F.dropout(z, p) with training=True always (graphsc.py:409)
        c.seed, c.step0 = _draw_seed(), 0
        c.bad, c.workspace, c.workspace_bytes = self.bad.data_ptr(), self.ws.data_ptr(), nbytes
        c.grads = self.grads.data_ptr() if self.grads is not None else None
        self._keep = (g.rowptr, g.col, g.val, feats)

    def run(self, seeds: torch.Tensor, n_steps: int, z_out: torch.Tensor, loss_out: torch.Tensor):
        lib = _lib.load()
        assert seeds.dtype == torch.int64 and seeds.is_contiguous() and seeds.numel() >= n_steps * self.batch
        assert z_out.is_contiguous() and z_out.shape == (n_steps * self.batch, self.emb_dim) and loss_out.numel() >= n_steps
        c = self.cfg
        self._hyper(c, self.optim)
        c.w1, c.b1, c.w2, c.b2 = (_adam_struct(self.optim, p) for p in self.params)  # (re-read every call: ``p.data = ...`` or a re-created state moves them)
        c.dropout = float(self.model.dropout.p) if self.model.dropout is not None and self.model.training else 0.0
        c.seeds, c.z_out, c.loss_out = seeds.data_ptr(), z_out.data_ptr(), loss_out.data_ptr()
        if self.world == 1:
            c.phase = 0
            self._run(lib.dh_graphsc_steps, "graphsc_steps", 0, n_steps)
        else:
            import torch.distributed as dist
            for s in range(n_steps):
                c.phase = 1
                self._run(lib.dh_graphsc_steps, "graphsc_steps", s, 1)
                dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)
                self.grads /= self.world
                c.phase = 2
                self._run(lib.dh_graphsc_steps, "graphsc_steps", s, 1)
        c.step0 += n_steps


    def aggregate(self, seeds: torch.Tensor) -> torch.Tensor:
        """[2, batch, in_feats]: what ``WeightedGraphConv`` multiplies by its weight for the batch ``seeds`` — sum_e w_e D_out^-1/2 x_src,
        scaled by D_in^-1/2 (and 1 / in-degree for "mean") — for the two forwards of a batch, each with its own draw of the input dropout
        (dh_graphsc_steps phase 3: two launches on the graph's CSR rows; no block, no gradient — the input features are a leaf)."""
        lib = _lib.load()
        assert seeds.dtype == torch.int64 and seeds.is_contiguous() and seeds.numel() == self.batch
        c = self.cfg
        out = torch.empty((2, self.batch, int(c.in_feats)), dtype=torch.float32, device=seeds.device)
        c.dropout = float(self.model.dropout.p) if self.model.dropout is not None and self.model.training else 0.0
        c.w1, c.b1, c.w2, c.b2 = (_adam_struct(self.optim, p) for p in self.params)
        c.seeds, c.ax_out, c.phase = seeds.data_ptr(), out.data_ptr(), 3
        self._run(lib.dh_graphsc_steps, "graphsc_aggregate", 0, 1)
        c.phase, c.ax_out = 0, None
        c.step0 += 1
        return out


class ScDeepSortStepper(_Stepper):
    """``ScDeepSort.cal_loss``'s batch loop (scdeepsort.py:233-257) on dh_scdeepsort_steps."""

    @staticmethod
    def eligible(model, graph, batch_size: int, optim, n_classes: int) -> bool:
        import torch.nn as nn

        from .autograd import HipLinear
        if not adam_plain(optim) or not hasattr(graph, "gene_prefix") or graph.gene_prefix() < 0 or graph.device.type != "cuda":
            return False
        if len(model.layers) != 1:
            return False
        layer = model.layers[0]
        drop, lin, act, norm = layer.layers
        if layer.use_neigh or type(lin) is not HipLinear or type(act) is not nn.ReLU or type(norm) is not nn.Identity or lin.bias is None or model.linear.bias is None:
            return False
        if not isinstance(drop, (nn.Dropout, nn.Identity)):
            return False
        feats = graph.ndata["features"]
        if feats.dtype not in (torch.float32, torch.bfloat16) or feats.dim() != 2 or feats.stride(1) != 1:
            return False
        return bool(_lib.load().dh_scdeepsort_step_supported(int(batch_size), lin.in_features, lin.out_features, int(n_classes)))

    def __init__(self, model, graph, batch_size: int, optim, world: int = 1):
        lib = _lib.load()
        _lib.require_device()
        self.model, self.optim, self.g, self.batch, self.world = model, optim, graph, int(batch_size), int(world)
        layer = model.layers[0]
        lin = layer.layers[1]
        self.params = [lin.weight, lin.bias, model.linear.weight, model.linear.bias]
        ensure_adam_state(optim, self.params)
        feats = graph.ndata["features"]
        d, h, ncls = lin.in_features, lin.out_features, model.linear.out_features
        dev = feats.device
        self.bad = torch.zeros(1, dtype=torch.int32, device=dev)
        nbytes = int(lib.dh_scdeepsort_step_workspace_bytes(self.batch, d, h, ncls))
        self.ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        self.grads = torch.empty(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev) if world > 1 else None
        self.neigh = torch.zeros((self.batch, d), dtype=torch.float32, device=dev) if layer.compute_neigh else None
        self.cell_id = graph.ndata["cell_id"].to(torch.int32).contiguous()
        self.labels = graph.ndata["label"].to(torch.int64).contiguous()
        self.alpha = model.alpha.detach().reshape(-1)  # a view: never updated (no gradient reaches it, gnn.py:92), read every step
        c = self.cfg = ScDeepSortStep()
        c.rowptr, c.col, c.val, c.features = graph.rowptr.data_ptr(), graph.col.data_ptr(), graph.val.data_ptr(), feats.data_ptr()
        c.ld_features, c.n_nodes, c.n_genes = feats.stride(0), graph.number_of_nodes(), graph.gene_prefix()
        c.cell_id, c.labels, c.alpha = self.cell_id.data_ptr(), self.labels.data_ptr(), self.alpha.data_ptr()
        c.batch, c.dim_in, c.hidden, c.n_classes = self.batch, d, h, ncls
        c.features_bf16 = 1 if feats.dtype == torch.bfloat16 else 0
        c.w1, c.b1, c.w2, c.b2 = (_adam_struct(optim, p) for p in self.params)
        self._hyper(c, optim)
        c.seed, c.step0 = _draw_seed(), 0
        c.neigh_out = self.neigh.data_ptr() if self.neigh is not None else None
        c.bad, c.workspace, c.workspace_bytes = self.bad.data_ptr(), self.ws.data_ptr(), nbytes
        c.grads = self.grads.data_ptr() if self.grads is not None else None
        self._keep = (graph.rowptr, graph.col, graph.val, feats)

    def _dropout_p(self) -> float:
        drop = self.model.layers[0].layers[0]
        return float(drop.p) if isinstance(drop, torch.nn.Dropout) and self.model.training else 0.0

    def run(self, seeds: torch.Tensor, n_steps: int, loss_out: torch.Tensor):
        lib = _lib.load()
        assert seeds.dtype == torch.int64 and seeds.is_contiguous() and seeds.numel() >= n_steps * self.batch and loss_out.numel() >= n_steps
        c = self.cfg
        self._hyper(c, self.optim)
        c.w1, c.b1, c.w2, c.b2 = (_adam_struct(self.optim, p) for p in self.params)  # (re-read every call, as above)
        c.alpha = self.model.alpha.detach().reshape(-1).data_ptr()
        c.dropout = self._dropout_p()
        c.seeds, c.loss_out = seeds.data_ptr(), loss_out.data_ptr()
        if self.world == 1:
            c.phase = 0
            self._run(lib.dh_scdeepsort_steps, "scdeepsort_steps", 0, n_steps)
        else:
            import torch.distributed as dist
            for s in range(n_steps):
                c.phase = 1
                self._run(lib.dh_scdeepsort_steps, "scdeepsort_steps", s, 1)
                dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)
                self.grads /= self.world
                c.phase = 2
                self._run(lib.dh_scdeepsort_steps, "scdeepsort_steps", s, 1)
        c.step0 += n_steps
        if self.neigh is not None and n_steps:
            self.model.layers[0].last_neigh = self.neigh  # the (discarded) aggregation of the last batch, as the eager layer keeps it


def dropout_mask(n: int, p: float, seed: int, step: int, sid: int, device) -> torch.Tensor:
    """The in-kernel dropout draw as a tensor (dh_ministep_dropout_mask_f32): 0 or 1 / (1 - p) per element."""
    from . import kernels
    lib = kernels._lib_ready()
    out = torch.empty(n, dtype=torch.float32, device=device)
    kernels._call("ministep_dropout_mask", lib.dh_ministep_dropout_mask_f32, int(n), float(p), int(seed), int(step), int(sid), out.data_ptr(), kernels._stream())
    return out
