"""scDeepSort on MI355X — drop-in for dance/modules/single_modality/cell_type_annotation/scdeepsort.py:26-349
(``GNN`` :26-88, ``ScDeepSort`` :91-349): same constructor / fit / predict / predict_proba / score signatures and
``state_dict`` keys (``alpha``, ``layers.{i}.alpha``, ``layers.{i}.layers.1.weight|bias``, ``linear.weight|bias``).

The training loop is the reference's (full-fan-out in-neighbour blocks of ``batch_size`` cells, Adam, summed CE);
the per-row ``.item()`` loop of ``evaluate`` (:278-283) is replaced by the equivalent vectorised device ops.
"""
import os
import time
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from .... import kernels
from ....autograd import CrossEntropySum, HipLinear
from ....cellgraph import DataLoader, NeighborSampler
from ....nn import AdaptiveSAGE
from ....transforms import Compose, SetConfig
from ....transforms.graph import PCACellFeatureGraph
from ...base import BaseClassificationMethod


class GNN(nn.Module):

    def __init__(self, dim_in: int, dim_out: int, dim_hid: int, n_layers: int, gene_num: int, activation=None, norm=None,
                 dropout: float = 0., *, compute_dtype: str = "fp32"):
        super().__init__()
        if compute_dtype not in ("fp32", "bf16"):
            raise ValueError(f"compute_dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        self.n_layers = n_layers
        self.gene_num = gene_num
        # [gene_num] is alpha of gene-gene self loop, [gene_num+1] is alpha of cell-cell self loop, the rest are betas
        self.alpha = nn.Parameter(torch.tensor([1] * (self.gene_num + 2), dtype=torch.float32).unsqueeze(-1))
        dropout_layer = nn.Dropout(p=dropout) if dropout > 0 else nn.Identity()
        act_layer = activation or nn.Identity()
        norm_layer = norm or nn.Identity()
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            self.layers.append(AdaptiveSAGE(dim_in if i == 0 else dim_hid, dim_hid, self.alpha, dropout_layer, act_layer, norm_layer))
        self.linear = HipLinear(dim_hid, dim_out)
        self.linear.out_dtype = torch.float32  # logits leave the bf16 path in fp32 (no effect on fp32 inputs)
        nn.init.xavier_uniform_(self.linear.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, blocks, x):
        assert len(blocks) == len(self.layers), f"Inonsistent layer info: {len(blocks)=} vs {len(self.layers)=}"
        if self.compute_dtype == "bf16" and x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)  # bf16 storage of node features and activations
        for block, layer in zip(blocks, self.layers):
            x = layer(block, x)
        return self.linear(x)


HIPGRAPH = os.environ.get("DANCE_AMD_HIPGRAPH", "1") != "0"
HIPGRAPH_MIN_BATCHES = int(os.environ.get("DANCE_AMD_HIPGRAPH_MIN_BATCHES", "64"))  # the capture (two eager steps + instantiation, tens of ms) must be amortised
HIPGRAPH_MAX_BATCH = int(os.environ.get("DANCE_AMD_HIPGRAPH_MAX_BATCH", "2048"))    # above this a step is kernel-bound: replaying gains little (see fit)
# The persistent step (dance_amd/ministep.py, csrc/ministep.hip): every full training batch of an epoch behind ONE C call, four launches
# per step (five with the reference's discarded aggregation), straight off the graph's CSR rows.  The default for the reference's model
# shape (one AdaptiveSAGE layer whose output ignores the aggregation + ReLU, Linear head) up to MINISTEP_MAX_BATCH cells per batch; above
# that the Linear layers want the large-tile / bf16 GEMMs of the general loop.  DANCE_AMD_MINISTEP=0: the captured / eager loop.
MINISTEP = os.environ.get("DANCE_AMD_MINISTEP", "1") != "0"
MINISTEP_MAX_BATCH = int(os.environ.get("DANCE_AMD_MINISTEP_MAX_BATCH", "4096"))


class ScDeepSort(BaseClassificationMethod):

    shuffle_generator = None  # host torch.Generator for a reproducible train/val split and batch order (see fit)
    # measurement aid (scripts/bench_configs.py): True -> ``epoch_ms`` holds the device time between the epoch boundaries of the last fit
    record_epoch_times = False
    # Evaluation / prediction passes (model.eval(), no gradient): per-cell logits do not depend on how the cells are batched,
    # so with one SAGE layer they are computed for ALL cells in one piece straight on the graph's CSR rows — no sampling, no
    # feature gathers — and indexed; the two evaluate() calls of an epoch share one such pass.  False = the reference's
    # batch-by-batch loop over sampled blocks (scdeepsort.py:272-283,299-330), kept for block-exact parity checks.
    full_graph_eval = True
    # Captured steps as two graphs with the gradient all-reduce between them (the form used with more than one process); True forces it
    # on one process too (tests)
    capture_split = False

    def __init__(self, dim_in: int, dim_hid: int, num_layers: int, species: str, tissue: str, *, dropout: int = 0,
                 batch_size: int = 500, device: str = "cuda", save_root=None, verbose: bool = True,
                 compute_dtype: str = "fp32"):
        # compute_dtype="bf16" (not in the reference, which is fp32 only): node features / activations are stored as
        # bf16 and the dense updates run on the bf16 matrix cores with fp32 accumulation (BASELINE.json config 3)
        if compute_dtype not in ("fp32", "bf16"):
            raise ValueError(f"compute_dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")
        if compute_dtype == "bf16" and (dim_in % 8 or dim_hid % 8):
            raise ValueError("compute_dtype='bf16' needs dim_in and dim_hid to be multiples of 8 (16-byte bf16 rows)")
        self.compute_dtype = compute_dtype
        self.dense_dim = dim_in
        self.hidden_dim = dim_hid
        self.n_layers = num_layers
        self.dropout = dropout
        self.species = species
        self.tissue = tissue
        self.batch_size = batch_size
        self.device = device
        self.verbose = verbose
        self.postfix = time.strftime("%d_%m_%Y") + "_" + time.strftime("%H:%M:%S")
        self.prj_path = Path(save_root).resolve() if save_root else Path().resolve()
        self.save_path = (self.prj_path / "saved_models" / "single_modality" / "cell_type_annotation" / "pretrained" /
                          self.species / "models")
        if not self.save_path.exists():
            self.save_path.mkdir(parents=True)

    @staticmethod
    def preprocessing_pipeline(n_components: int = 400, log_level="INFO"):
        return Compose(
            PCACellFeatureGraph(n_components=n_components, split_name="train"),
            SetConfig({"label_channel": "cell_type"}),
            log_level=log_level,
        )

    def _typed(self, graph):
        """bf16 mode: a shallow copy of the device graph whose ``features`` are stored once as bf16 (the caller's
        graph keeps its fp32 features)."""
        if self.compute_dtype != "bf16" or graph.ndata["features"].dtype == torch.bfloat16:
            return graph
        return graph.with_ndata(features=graph.ndata["features"].to(torch.bfloat16))

    def _print(self, *a):
        if self.verbose:
            print(*a)

    def fit(self, graph, labels: torch.Tensor, epochs: int = 300, lr: float = 1e-3, weight_decay: float = 0,
            val_ratio: float = 0.2):
        with kernels.mini_batch_products():  # the step's small fp32 products on dh_gemm_f32_small (a captured step records them so)
            return self._fit(graph, labels, epochs, lr, weight_decay, val_ratio)

    def _fit(self, graph, labels, epochs, lr, weight_decay, val_ratio):
        gene_mask = graph.ndata["cell_id"] != -1
        cell_mask = graph.ndata["cell_id"] == -1
        num_genes = int(gene_mask.sum())
        num_cells = int(cell_mask.sum())
        labels = torch.as_tensor(labels)
        self.num_labels = labels.max().item() + 1

        # scdeepsort.py:157-160.  Drawn on the device by default; with ``self.shuffle_generator`` (a host torch.Generator)
        # the split and every epoch's batch order come from that generator, as the reference draws them from the CPU RNG
        gen = self.shuffle_generator
        perm = (torch.randperm(num_cells, device=self.device) if gen is None
                else torch.randperm(num_cells, generator=gen).to(self.device)) + num_genes
        from .... import sharding
        if sharding.world_info()[1] > 1:  # data parallel: one train / validation split for all ranks
            torch.distributed.broadcast(perm, src=0)
        num_val = int(num_cells * val_ratio)
        val_idx = perm[:num_val]
        train_idx = perm[num_val:]

        full_labels = -torch.ones(num_genes + num_cells, dtype=torch.long)
        full_labels[-num_cells:] = labels.cpu()
        graph = graph.to(self.device)
        graph.ndata["label"] = full_labels.to(self.device)

        graph = self._typed(graph)
        self.model = GNN(self.dense_dim, self.num_labels, self.hidden_dim, self.n_layers, num_genes, activation=nn.ReLU(),
                         dropout=self.dropout, compute_dtype=self.compute_dtype).to(self.device)
        self.sampler = NeighborSampler(fanouts=[-1] * self.n_layers, edge_dir="in")
        # One captured hipGraph per training step (static-shape block of the batch's cells, forward, loss, backward, Adam): at the
        # reference's batch size (500) a step is a few dozen microsecond kernels and the loop is launch-bound.  One layer,
        # CellFeatureGraph node layout, enough full batches (of this rank's share of the cells) to amortise the capture;
        # DANCE_AMD_HIPGRAPH=0 keeps the eager loop.  More than one process: two graphs per step, the gradient all-reduce between them.
        self._world = sharding.world_info()[1]
        n_full = -(-len(train_idx) // self._world) // self.batch_size
        self._use_graph = (HIPGRAPH and self.n_layers == 1 and graph.gene_prefix() >= 0
                           and str(self.device).startswith("cuda") and n_full * max(int(epochs), 1) >= HIPGRAPH_MIN_BATCHES and 1 < self.batch_size <= HIPGRAPH_MAX_BATCH
                           and not any(layer.use_neigh for layer in self.model.layers))
        self._captured = None
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay, capturable=self._use_graph,
                                          fused=str(self.device).startswith("cuda"))  # one kernel per step instead of ~14
        self.loss_fn = CrossEntropySum()  # nn.CrossEntropyLoss(reduction="sum") of :185 as one kernel (torch's nll_loss reductions are one workgroup)
        from ....ministep import ScDeepSortStepper
        self._stepper = None
        self._use_mini = (MINISTEP and not self.capture_split and str(self.device).startswith("cuda") and n_full >= 1 and self.batch_size <= MINISTEP_MAX_BATCH
                          and ScDeepSortStepper.eligible(self.model, graph, self.batch_size, self.optimizer, self.num_labels))
        if self._use_mini:
            self._use_graph = False

        # more than one process: data parallelism over the training cells (the graph is replicated, the model is small):
        # every rank trains on its share, gradients are averaged with one flat all-reduce per step (dance_amd/sharding.py)
        if self._world > 1:
            sharding.broadcast_parameters(self.model)
        self._print(f"Train Number: {len(train_idx)}, Val Number: {len(val_idx)}")
        max_val_acc, _train_acc, _epoch = 0, 0, 0
        final_val_correct_num = final_val_unsure_num = 0
        best_state_dict = None
        marks = []
        for epoch in range(epochs):
            if self.record_epoch_times and torch.cuda.is_available():  # measurement aid: see the class attribute
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
            loss = self.cal_loss(graph, train_idx)
            logits = self._full_graph_logits(graph)  # None: block-by-block evaluation
            train_acc = self.evaluate(graph, train_idx, _logits=logits)[-1]
            val_correct, val_unsure, val_acc = self.evaluate(graph, val_idx, _logits=logits) if len(val_idx) else (0, 0, 0.0)
            if max_val_acc <= val_acc:
                final_val_correct_num, final_val_unsure_num = val_correct, val_unsure
                _train_acc, _epoch, max_val_acc = train_acc, epoch, val_acc
                self.save_model()
                best_state_dict = deepcopy(self.model.state_dict())
            self._print(f">>>>Epoch {epoch:04d}: Train Acc {train_acc:.4f}, Loss {loss / len(train_idx):.4f}, "
                        f"Val correct {val_correct}, Val unsure {val_unsure}, Val Acc {val_acc:.4f}")
        if marks:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
            marks[-1].synchronize()
            self.epoch_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        if best_state_dict is not None:
            self.model.load_state_dict(best_state_dict)
        self._print(f"---{self.species} {self.tissue} Best val result:---")
        self._print(f"Epoch {_epoch:04d}, Train Acc {_train_acc:.4f}, Val Correct Num {final_val_correct_num}, "
                    f"Val Total Num {len(val_idx)}, Val Unsure Num {final_val_unsure_num}")

    def _captured_forward_backward(self, block):
        blk = block.rebuild()
        loss = self.loss_fn(self.model([blk], blk.srcdata["features"]), blk.dstdata["label"])
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        return loss.detach()

    def _captured_optimiser_step(self):
        if not kernels.adam_step(self.optimizer):  # dh_adam_step_f32 once the optimiser's state exists
            self.optimizer.step()

    def _captured_step_body(self, block):
        loss = self._captured_forward_backward(block)
        self._captured_optimiser_step()
        return loss

    def _capture_step(self, graph, first_seeds):
        """Record one training step on a StaticCellBlock as a hipGraph (dance_amd/capture.py; with more than one process: two graphs
        and the gradient all-reduce between them); the model and the (fresh) optimiser state are put back to where they were, so
        capturing is not a training step."""
        from .... import sharding
        from ....capture import CapturedStep
        from ....cellgraph import StaticCellBlock
        if self.optimizer.state:
            raise RuntimeError("capture expects a fresh optimiser")
        block = StaticCellBlock(graph, self.batch_size)
        saved = deepcopy(self.model.state_dict())
        block.seeds.copy_(first_seeds)
        g = CapturedStep(lambda: self._captured_forward_backward(block), self._captured_optimiser_step, first_seeds.device,
                         split=getattr(self, "_world", 1) > 1 or self.capture_split, between=lambda: sharding.allreduce_gradients(self.model),
                         keep_alive=lambda: [p.grad for p in self.model.parameters() if p.grad is not None], params=list(self.model.parameters()))
        loss = g.outputs
        self.model.load_state_dict(saved)       # in place: the graph keeps pointing at these tensors
        for st in self.optimizer.state.values():  # moments and step counters back to zero, in place
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        block.bad.zero_()
        return g, block, loss

    def cal_loss(self, graph, idx: torch.Tensor):
        self.model.train()
        losses, sizes = [], []
        from .... import sharding
        if getattr(self, "_world", 1) > 1:
            idx = sharding.shard_seed_ids(idx)
        if getattr(self, "_use_mini", False) and idx.numel() // self.batch_size >= 1:
            from ....ministep import ScDeepSortStepper
            idx = idx.to(self.device)
            perm = (torch.randperm(idx.numel(), device=idx.device) if self.shuffle_generator is None else
                    torch.randperm(idx.numel(), generator=self.shuffle_generator).to(idx.device))  # the loader's own order
            idx = idx[perm].contiguous()
            n_full = idx.numel() // self.batch_size
            if self._stepper is None or self._stepper.g is not graph:
                self._stepper = ScDeepSortStepper(self.model, graph, self.batch_size, self.optimizer, getattr(self, "_world", 1))
            loss_all = torch.empty(n_full, dtype=torch.float32, device=idx.device)
            self._stepper.run(idx, n_full, loss_all)  # all full batches of the epoch: one C call
            self._stepper.check_flags("ScDeepSort.fit", mask=5)  # one read per epoch
            losses, sizes = list(loss_all.unbind(0)), [self.batch_size] * n_full
            tail = idx[n_full * self.batch_size:]
            batches = [self.sampler.sample(graph, tail, True)] if tail.numel() else []
        elif getattr(self, "_use_graph", False) and idx.numel() // self.batch_size >= 1:
            idx = idx.to(self.device)
            perm = (torch.randperm(idx.numel(), device=idx.device) if self.shuffle_generator is None else
                    torch.randperm(idx.numel(), generator=self.shuffle_generator).to(idx.device))  # the loader's own order
            idx = idx[perm]
            n_full = idx.numel() // self.batch_size
            if self._captured is None:
                self._captured = self._capture_step(graph, idx[:self.batch_size])
            cg, block, static_loss = self._captured
            loss_all = torch.empty(n_full, dtype=torch.float32, device=idx.device)
            for i in range(n_full):
                block.seeds.copy_(idx[i * self.batch_size:(i + 1) * self.batch_size])
                cg.replay()
                loss_all[i].copy_(static_loss)
            if int(block.bad) & 1:  # (bit 2, "no single self loop", only matters to graph-sc's identity decoder target)
                raise RuntimeError("ScDeepSort.fit: a training seed has a non-gene in-neighbour other than itself, or a batch has more in-edges "
                                   "than the static block holds — not a CellFeatureGraph-layout graph (set DANCE_AMD_HIPGRAPH=0 for the eager loop)")
            losses, sizes = list(loss_all.unbind(0)), [self.batch_size] * n_full
            tail = idx[n_full * self.batch_size:]
            batches = [self.sampler.sample(graph, tail, True)] if tail.numel() else []
        else:
            batches = DataLoader(graph=graph, indices=idx, sampler=self.sampler, batch_size=self.batch_size, shuffle=True,
                                 generator=self.shuffle_generator)
        for _, _, blocks in batches:
            input_features = blocks[0].srcdata["features"]
            output_labels = blocks[-1].dstdata["label"]
            output_predictions = self.model(blocks, input_features)
            loss = self.loss_fn(output_predictions, output_labels)
            self.optimizer.zero_grad(set_to_none=getattr(self, "_captured", None) is None)  # (see dance_amd/capture.py: next to a captured step
            loss.backward()                                                                # the gradients stay the graph's tensors)
            sharding.allreduce_gradients(self.model)
            self.optimizer.step()
            sizes.append(blocks[-1].num_dst_nodes())
            losses.append(loss.detach())  # read back once per epoch, not once per batch (scdeepsort.py:247-248)
        total = sum(v * n for v, n in zip(torch.stack(losses).tolist(), sizes)) if losses else 0.0
        return total / max(sum(sizes), 1)

    @torch.no_grad()
    def _full_graph_logits(self, graph):
        """Logits of every cell from ONE pass over the graph (see ``full_graph_eval``); None when that mode does not apply (a node
        order other than CellFeatureGraph's genes-first).  With L layers the inner L - 1 layers update EVERY node (what the
        sampler's fanout -1 blocks reach from any cell: all its genes, then all their cells, scdeepsort.py:183) — their gene rows
        aggregate ~1e5 cells each through the split-K matrix-core kernel — and the last layer the cell rows."""
        if not self.full_graph_eval or graph.gene_prefix() < 0:
            return None
        self.model.eval()
        blk = graph.cell_rows_block()
        inner = [graph.all_rows_block()] * (self.n_layers - 1)
        return self.model(inner + [blk], blk.srcdata["features"])

    @torch.no_grad()
    def evaluate(self, graph, idx: torch.Tensor, unsure_rate: float = 2.0, *, _logits=None):
        with kernels.mini_batch_products():  # the same product kernels inside and outside fit: logits are bit-reproducible
            return self._evaluate(graph, idx, unsure_rate, _logits)

    def _evaluate(self, graph, idx, unsure_rate, _logits):
        self.model.eval()
        if _logits is None:
            _logits = self._full_graph_logits(graph)
        if _logits is not None:
            rows = idx.to(_logits.device) - graph.gene_prefix()
            pred, labels = _logits[rows], graph.ndata["label"][idx.to(_logits.device)]
            unsure = pred.max(1).values < unsure_rate / self.num_labels  # :280-281 (on raw logits, as written)
            stats = torch.stack((((pred.argmax(1) == labels) & ~unsure).sum(), unsure.sum())).tolist()
            return stats[0], stats[1], stats[0] / len(idx)
        total_correct = total_unsure = 0
        dataloader = DataLoader(graph=graph, indices=idx, sampler=self.sampler, batch_size=self.batch_size, shuffle=True,
                                generator=self.shuffle_generator)
        for _, _, blocks in dataloader:
            input_features = blocks[0].srcdata["features"]
            output_labels = blocks[-1].dstdata["label"]
            pred = self.model(blocks, input_features)
            unsure = pred.max(1).values < unsure_rate / self.num_labels  # :280-281 (on raw logits, as written)
            total_unsure += int(unsure.sum())
            total_correct += int(((pred.argmax(1) == output_labels) & ~unsure).sum())
        return total_correct, total_unsure, total_correct / len(idx)

    def save_model(self):
        state = {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict()}
        torch.save(state, self.save_path / f"{self.species}-{self.tissue}.pt")

    def load_model(self):
        filename = f"{self.species}-{self.tissue}.pt"
        model_path = self.prj_path / "pretrained" / self.species / "models" / filename
        state = torch.load(model_path, map_location=self.device)
        self.model.load_state_dict(state["model"])

    @torch.no_grad()
    def predict_proba(self, graph):
        with kernels.mini_batch_products():  # as in fit (see evaluate)
            return self._predict_proba(graph)

    def _predict_proba(self, graph):
        self.model.eval()
        cell_mask = (graph.ndata["cell_id"] == -1).cpu()
        idx = torch.where(cell_mask)[0]
        graph = self._typed(graph.to(self.device))
        full = self._full_graph_logits(graph)
        if full is not None:
            return nn.functional.softmax(full.float(), dim=-1).cpu().numpy()
        logits = torch.zeros(graph.number_of_nodes(), self.num_labels)
        dataloader = DataLoader(graph=graph, indices=idx, sampler=self.sampler, batch_size=self.batch_size)
        for _, output_nodes, blocks in dataloader:
            input_features = blocks[0].srcdata["features"]
            logits[output_nodes.cpu()] = self.model(blocks, input_features).detach().cpu()
        return nn.functional.softmax(logits[cell_mask], dim=-1).numpy()

    def predict(self, graph, unsure_rate: float = 2.0, return_unsure: bool = False):
        pred_prob = self.predict_proba(graph)
        pred = pred_prob.argmax(1)
        unsure = pred_prob.max(1) < unsure_rate / self.num_labels
        return (pred, unsure) if return_unsure else pred
