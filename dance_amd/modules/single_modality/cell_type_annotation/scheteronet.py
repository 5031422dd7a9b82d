"""scHeteroNet on MI355X — drop-in for the GNN part of
dance/modules/single_modality/cell_type_annotation/scheteronet.py (``MLP`` :339-371, ``HetConv`` :374-386,
``ZINBDecoder`` :389-442, ``HeteroNet`` :465-573, ``scHeteroNet`` :576-789, ``ZINBLoss`` :289-336, ``contrastive_loss``
:281-286): same constructors, parameter names (``state_dict`` keys: ``encoder.feature_embed.lins.*``, ``encoder.bns.*``,
``encoder.final_project.*``, ``encoder.ZINB.*``) and method signatures.

What runs where:

* ``HeteroNet.init_adj`` (:507-539) — the reference builds the one-hop and the "two-hop minus one-hop" adjacencies with
  torch_sparse + scipy on the host and keeps them as SparseTensors.  Here both are device CSR patterns: the one-hop CSR by
  destination comes straight from the edge list, the two-hop pattern from a symbolic SpGEMM on the device
  (dh_csr_two_hop_*: path keys, radix sort, run heads); ``gcn_norm(add_self_loops=False)`` is not materialised — its
  D^-1/2 factors ride the SpMM as row / column scales.
* ``HetConv.forward`` (:383-386; the reference moves x to the CPU for torch_sparse.matmul) = two fused CSR SpMM launches
  (dh_spmm_csr_f32 with rowscale / colscale) + the concat; backward = the same kernel on the cached transposes.
* ``propagation`` / ``two_hop_propagation`` (:611-640) = mean-aggregation SpMVs of the same kernel ((adj @ adj) e is
  evaluated as adj (adj e)).
* MLP / decoder / final projection: ``HipLinear`` (dh_gemm_f32 on the matrix cores); BatchNorm, dropout, the ZINB and
  contrastive losses are torch elementwise ops, as in the reference.

PARITY NOTE.  ``adj_t.remove_diag(0)`` at :522 and :524 discards its result (torch_sparse's ``remove_diag`` returns a new
tensor), so the reference's adjacencies KEEP their self loops (HeteronetGraph lists every cell as its own first neighbour)
despite the docstring; ``adj_t2 = (A A - A > 0)`` is evaluated on those.  ``remove_self_loops=False`` (the default)
reproduces the code as written; ``True`` gives what the docstring says.
"""
import logging
from functools import partial
from typing import Any, Mapping, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import kernels
from ....autograd import HipLinear, linear, spmm, zinb_nll
from ....graph import CSRGraph, TensorKeyedCache
from ....transforms import Compose, SetConfig
from ....transforms.graph import HeteronetGraph
from ...base import BaseClassificationMethod

logger = logging.getLogger("dance")


def eval_acc(true_labels, model_output, acc):
    """scheteronet.py:44-65: one-hot the labels and score the argmax predictions with ``acc``."""
    predicted = torch.argmax(model_output, dim=1)
    if true_labels.ndim == 2 and true_labels.shape[1] == 1:
        true_labels = true_labels.squeeze(1)
    return acc(F.one_hot(true_labels.long(), num_classes=model_output.shape[1]), predicted)


class NCDataset:
    """scheteronet.py:68-107: ``graph`` = {edge_index, edge_feat, node_feat, num_nodes}, ``label``; attribute access to
    ``x`` / ``edge_index`` / ``y`` / ``splits`` / ``node_idx`` as the model methods use them."""

    def __init__(self, name):
        self.name = name
        self.graph = {}
        self.label = None

    def __getitem__(self, idx):
        assert idx == 0, "This dataset has only one graph"
        return self.graph, self.label

    def __len__(self):
        return 1

    def __repr__(self):
        return f"{self.__class__.__name__}({len(self)})"


def contrastive_loss(z1, z2, temperature=0.5):
    z1 = F.normalize(z1, dim=-1)
    z2 = F.normalize(z2, dim=-1)
    logits = linear(z1, z2) / temperature   # z1 z2^T on dh_gemm_f32, gradients to both operands (autograd._LinearFn)
    labels = torch.arange(z1.size(0), device=z1.device)
    return F.cross_entropy(logits, labels)


class ZINBLoss(nn.Module):
    """scheteronet.py:289-336 (the same ZINB NLL as dance/utils/loss.py:780-829) on the fused kernels (autograd.zinb_nll)."""

    def forward(self, x, mean, disp, pi, scale_factor, ridge_lambda=0.0):
        return zinb_nll(x, mean, disp, pi, scale_factor, ridge_lambda)


class MLP(nn.Module):
    # reference: dance/modules/single_modality/cell_type_annotation/scheteronet.py:339-371 — module skeleton transcribed (attribute names =
    # state_dict keys; layer order); the Linear layers are HipLinear (dh_gemm_f32 with the ReLU in the epilogue)

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout=.5):
        super().__init__()
        self.lins = nn.ModuleList()
        self.bns = nn.ModuleList()
        if num_layers == 1:
            self.lins.append(HipLinear(in_channels, out_channels))
        else:
            self.lins.append(HipLinear(in_channels, hidden_channels))
            self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(HipLinear(hidden_channels, hidden_channels))
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(HipLinear(hidden_channels, out_channels))
        self.dropout = dropout

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    def forward(self, x, edge_index=None):
        for i, lin in enumerate(self.lins[:-1]):
            x = lin(x, fuse_relu=True)
            x = self.bns[i](x)
            x = F.dropout(x, p=self.dropout, training=self.training)
        return self.lins[-1](x)


class NormAdj:
    """A 0/1 adjacency pattern (CSR by destination, on the device) with its gcn_norm factors D^-1/2 (row sums; 0 for an
    empty row, as ``deg_inv_sqrt.masked_fill_(inf, 0)``): ``matmul(gcn_norm(A), x)`` = rowscale * (A (colscale * x))."""

    def __init__(self, rowptr, col, n):
        self.graph = CSRGraph(rowptr, col, None, n, n)
        deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32)
        self.dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))

    def matmul(self, x):
        return spmm(x, self.graph, rowscale=self.dis, colscale=self.dis)

    def to_dense(self):
        a = torch.zeros((self.graph.n_rows, self.graph.n_cols), device=self.dis.device)
        rows = torch.repeat_interleave(torch.arange(self.graph.n_rows, device=a.device),
                                       (self.graph.rowptr[1:] - self.graph.rowptr[:-1]).long())
        a[rows, self.graph.col.long()] = 1.0
        return self.dis[:, None] * a * self.dis[None, :]


class HetConv(nn.Module):
    """Neighborhood aggregation step: [A1_norm x | A2_norm x]."""

    def reset_parameters(self):
        pass

    def forward(self, x, adj_t, adj_t2):
        return torch.cat([adj_t.matmul(x), adj_t2.matmul(x)], dim=1)


class MeanAct(nn.Module):

    def forward(self, x):
        return torch.clamp(torch.exp(x), min=1e-5, max=1e6)


class DispAct(nn.Module):

    def forward(self, x):
        return torch.clamp(F.softplus(x), min=1e-4, max=1e4)


class ZINBDecoder(nn.Module):

    def __init__(self, input_dim, n_z, n_dec_1=128, n_dec_2=256, n_dec_3=512):
        super().__init__()
        self.n_dec_3 = n_dec_3
        self.input_dim = input_dim
        self.dec_1 = HipLinear(n_z, n_dec_1)
        self.dec_2 = HipLinear(n_dec_1, n_dec_3)
        self.dec_3 = HipLinear(n_dec_2, n_dec_3)  # unused in forward, as in the reference (:413,:438); kept for the state_dict
        self.dec_mean = nn.Sequential(HipLinear(self.n_dec_3, self.input_dim), MeanAct())
        self.dec_disp = nn.Sequential(HipLinear(self.n_dec_3, self.input_dim), DispAct())
        self.dec_pi = nn.Sequential(HipLinear(self.n_dec_3, self.input_dim), nn.Sigmoid())

    def forward(self, z):
        dec_h1 = self.dec_1(z, fuse_relu=True)
        dec_h3 = self.dec_2(dec_h1, fuse_relu=True)
        return self.dec_mean(dec_h3), self.dec_disp(dec_h3), self.dec_pi(dec_h3)


def edge_index_to_csr(edge_index, n, device):
    """``SparseTensor(row=col, col=row)`` of scheteronet.py:520: CSR by DESTINATION (row = edge target), ascending
    duplicate-free columns, 0/1 pattern."""
    row, col = edge_index[0].to(device).long(), edge_index[1].to(device).long()
    key = torch.unique(col * n + row)  # (dst, src) sorted
    dst, src = key // n, key % n
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
    return rowptr.to(torch.int32), src.to(torch.int32).contiguous()


class HeteroNet(nn.Module):

    def __init__(self, in_channels, hidden_channels, out_channels, edge_index, num_nodes, num_layers=2, dropout=0.5,
                 num_mlp_layers=1, use_bn=True, conv_dropout=True, dec_dim=[], device="cuda", remove_self_loops=False):
        super().__init__()
        self.feature_embed = MLP(in_channels, hidden_channels, hidden_channels, num_layers=num_mlp_layers, dropout=dropout)
        self.convs = nn.ModuleList()
        self.convs.append(HetConv())
        self.bns = nn.ModuleList()
        self.bns.append(nn.BatchNorm1d(hidden_channels * 2 * len(self.convs)))
        for l in range(num_layers - 1):
            self.convs.append(HetConv())
            if l != num_layers - 2:
                self.bns.append(nn.BatchNorm1d(hidden_channels * 2 * len(self.convs)))
        self.dropout = dropout
        self.activation = F.relu
        self.use_bn = use_bn
        self.conv_dropout = conv_dropout
        last_dim = hidden_channels * (2**(num_layers + 1) - 1)
        self.final_project = HipLinear(last_dim, out_channels)
        self.num_nodes = num_nodes
        self.device = device
        self.remove_self_loops = remove_self_loops
        self.init_adj(edge_index)
        self.ZINB = ZINBDecoder(in_channels, last_dim, n_dec_1=dec_dim[0], n_dec_2=dec_dim[1], n_dec_3=dec_dim[2])

    def reset_parameters(self):
        self.feature_embed.reset_parameters()
        self.final_project.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    def init_adj(self, edge_index):
        """Normalised one-hop adjacency and normalised ((A A) - A > 0) adjacency (scheteronet.py:507-539)."""
        n = self.num_nodes
        rowptr, col = edge_index_to_csr(edge_index, n, self.device)
        if self.remove_self_loops:
            rows = torch.repeat_interleave(torch.arange(n, device=col.device), (rowptr[1:] - rowptr[:-1]).long())
            keep = rows != col.long()
            rowptr = torch.zeros(n + 1, dtype=torch.int64, device=col.device)
            rowptr[1:] = torch.cumsum(torch.bincount(rows[keep], minlength=n), 0)
            rowptr, col = rowptr.to(torch.int32), col[keep].contiguous()
        rowptr2, col2 = kernels.csr_two_hop(rowptr, col, drop_diag=self.remove_self_loops)
        self.adj_t = NormAdj(rowptr, col, n)
        self.adj_t2 = NormAdj(rowptr2, col2, n)

    def forward(self, x, edge_index, decoder=False, save_path=None):
        # reference: scheteronet.py:541-566 — the layer sequence transcribed (embed, H2GCN convs with the running concatenation, jump, final
        # projection); every conv is two HIP SpMMs on the normalised one- and two-hop patterns
        adj_t, adj_t2 = self.adj_t, self.adj_t2
        x = self.feature_embed(x)
        x = self.activation(x)
        xs = [x]
        if self.conv_dropout:
            x = F.dropout(x, p=self.dropout, training=self.training)
        for i, conv in enumerate(self.convs[:-1]):
            x = conv(x, adj_t, adj_t2)
            if self.use_bn:
                x = self.bns[i](x)
            xs.append(x)
            if self.conv_dropout:
                x = F.dropout(x, p=self.dropout, training=self.training)
        x = self.convs[-1](x, adj_t, adj_t2)
        if self.conv_dropout:
            x = F.dropout(x, p=self.dropout, training=self.training)
        xs.append(x)
        x = torch.cat(xs, dim=-1)  # JumpingKnowledge('cat')
        if not self.conv_dropout:
            x = F.dropout(x, p=self.dropout, training=self.training)
        if save_path is not None:
            torch.save(x, save_path + "_embeddings.pt")
        if decoder:
            _mean, _disp, _pi = self.ZINB(x)
            return self.final_project(x), _mean, _disp, _pi
        return self.final_project(x)


def _mean_in_adj(edge_index, n, device):
    """``SparseTensor(row=col, col=row, value=1/d[col])`` of :616-621: out[c] = mean of e over the in-edges of c (edge
    multiplicities kept) — CSR by destination with the SpMM's mean reduction."""
    row, col = edge_index[0].to(device).long(), edge_index[1].to(device).long()
    order = torch.argsort(col * n + row)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(col, minlength=n), 0)
    return CSRGraph(rowptr.to(torch.int32), row[order].to(torch.int32).contiguous(), None, n, n)


class scHeteroNet(nn.Module, BaseClassificationMethod):

    def __init__(self, d, c, edge_index, num_nodes, hidden_channels, num_layers, dropout, use_bn, device, min_loss):
        super().__init__()
        self.device = device
        self.encoder = HeteroNet(d, hidden_channels, c, edge_index=edge_index, num_nodes=num_nodes, num_layers=num_layers,
                                 dropout=dropout, use_bn=use_bn, dec_dim=[32, 64, 128], device=device)
        self.encoder.to(device)
        self.to(device)
        self.min_loss = min_loss
        self._prop_cache = TensorKeyedCache()

    def reset_parameters(self):
        self.encoder.reset_parameters()

    @staticmethod
    def preprocessing_pipeline(log_level="INFO"):
        """scheteronet.py:592-604 with the count-matrix steps on the device: drop rare cell types, filter genes / cells, cell_ranger HVG
        (4000 genes, subset), SaveRaw, normalize_total, size factors, log1p, then the graph."""
        from ....transforms import (FilterCellsScanpy, FilterCellsType, FilterGenesScanpy, HighlyVariableGenesLogarithmizedByTopGenes, Log1P,
                                    NormalizeTotal, SaveRaw, UpdateSizeFactors)
        return Compose(
            FilterCellsType(),
            FilterGenesScanpy(min_counts=3),
            FilterCellsScanpy(min_counts=1),
            HighlyVariableGenesLogarithmizedByTopGenes(n_top_genes=4000, flavor="cell_ranger"),
            SaveRaw(),
            NormalizeTotal(),
            UpdateSizeFactors(),
            Log1P(),
            HeteronetGraph(),
            SetConfig({"label_channel": "cell_type"}),
            log_level=log_level,
        )

    def forward(self, dataset, save_path=None):
        x, edge_index = dataset.x.to(self.device), dataset.edge_index.to(self.device)
        return self.encoder(x, edge_index, save_path=save_path)

    def _prop_graph(self, edge_index, n):
        g = self._prop_cache.get(edge_index, n)
        if g is None:
            g = self._prop_cache.put(edge_index, _mean_in_adj(edge_index, n, self.device), n)
        return g

    def propagation(self, e, edge_index, prop_layers=1, alpha=0.5):
        """Energy belief propagation (scheteronet.py:611-623): e <- alpha e + (1 - alpha) mean_in(e)."""
        g = self._prop_graph(edge_index, e.shape[0])
        e = e.unsqueeze(1)
        for _ in range(prop_layers):
            e = e * alpha + spmm(e, g, reduce=kernels.REDUCE_MEAN) * (1 - alpha)
        return e.squeeze(1)

    def two_hop_propagation(self, e, edge_index, prop_layers=1, alpha=0.5):
        """:625-640: the two-hop operator adj @ adj applied as two mean-aggregation SpMVs."""
        g = self._prop_graph(edge_index, e.shape[0])
        e = e.unsqueeze(1)
        for _ in range(prop_layers):
            e = e * alpha + spmm(spmm(e, g, reduce=kernels.REDUCE_MEAN), g, reduce=kernels.REDUCE_MEAN) * (1 - alpha)
        return e.squeeze(1)

    def detect(self, dataset, node_idx, device, T, use_prop, use_2hop, oodprop, oodalpha):
        x, edge_index = dataset.x.to(device), dataset.edge_index.to(device)
        logits = self.encoder(x, edge_index)
        if dataset in ("proteins", "ppi"):
            logits = torch.stack([logits, torch.zeros_like(logits)], dim=2)
            neg_energy = T * torch.logsumexp(logits / T, dim=-1).sum(dim=1)
        else:
            neg_energy = T * torch.logsumexp(logits / T, dim=-1)
        if use_prop:
            prop = self.two_hop_propagation if use_2hop else self.propagation
            neg_energy = prop(neg_energy, edge_index, oodprop, oodalpha)
        return neg_energy[node_idx]

    def loss_compute(self, dataset_ind, dataset_ood, criterion, device, use_zinb):
        x_out, edge_index_out = dataset_ood.x.to(device), dataset_ood.edge_index.to(device)
        train_in_idx = dataset_ind.splits["train"]
        logits_in, _mean, _disp, _pi = (i[train_in_idx] for i in self.encoder(dataset_ind.x.to(device), dataset_ind.edge_index.to(device),
                                                                               decoder=use_zinb))
        self.encoder(x_out, edge_index_out)  # logits_out: computed and unused, as in the reference (:667)
        pred_in = F.log_softmax(logits_in, dim=1)
        loss = criterion(pred_in, (dataset_ind.y.to(device))[train_in_idx].squeeze(1))
        return loss, _mean, _disp, _pi, train_in_idx, logits_in

    def fit(self, dataset_ind, dataset_ood_tr, use_zinb, adata, zinb_weight, cl_weight, mask_ratio, criterion, optimizer):
        self.train()
        optimizer.zero_grad()
        loss, _mean, _disp, _pi, train_idx, logit_in = self.loss_compute(dataset_ind, dataset_ood_tr, criterion, self.device, use_zinb)
        if use_zinb:
            import scipy.sparse
            x_raw = adata.raw.X
            if isinstance(getattr(x_raw, "__dict__", {}).get("tensor"), torch.Tensor):  # DeviceArray left by the device pipeline: no host trip
                x_raw = x_raw.tensor.to(device=self.device, dtype=torch.float32)[train_idx]
            else:
                if scipy.sparse.issparse(x_raw):
                    x_raw = x_raw.toarray()
                x_raw = torch.as_tensor(np.asarray(x_raw), dtype=torch.float32).to(self.device)[train_idx]
            size_factors = torch.as_tensor(np.asarray(adata.obs["size_factors"]), device=self.device)[train_idx]
            loss = loss + zinb_weight * ZINBLoss()(x_raw, _mean, _disp, _pi, size_factors)
        if cl_weight != 0:
            X = dataset_ind.x.to(self.device)
            mask1 = (torch.rand_like(X) > mask_ratio).float()
            z1 = self.encoder(X * mask1, dataset_ind.edge_index.to(self.device))
            loss = loss + cl_weight * contrastive_loss(logit_in, z1)
        loss.backward()
        optimizer.step()
        return loss

    def predict_proba(self, dataset_ind, save_path=None):
        self.eval()
        with torch.no_grad():
            return F.softmax(self(dataset_ind, save_path=save_path), dim=1).cpu()

    def predict(self, dataset_ind):
        return torch.argmax(self.predict_proba(dataset_ind), dim=1)

    def score(self, x, y, idx, *, score_func: Optional[Union[str, Mapping[Any, float]]] = None, return_pred: bool = False,
              save_path=None, batch=None) -> Union[float, Tuple[float, Any]]:
        from ...base import resolve_score_func
        y_pred = self.predict_proba(x, save_path=save_path)
        if save_path is not None and batch is not None:
            np.savez_compressed(save_path + "_predictions.npz", predictions=y.cpu().numpy(), batch=batch)
        func = partial(eval_acc, acc=resolve_score_func(score_func or self._DEFAULT_METRIC))
        score = func(y[idx], y_pred[idx])
        return (score, y_pred) if return_pred else score

    def evaluate(self, dataset_ind, dataset_ood_te, criterion, eval_func, display_step, run, results, epoch, loss, dataset, T, use_prop,
                 use_2hop, oodprop, oodalpha):
        """scheteronet.py:714-734: one evaluation round of the training script — OOD detection measures + test score +
        validation loss appended to ``results[run]``; ``min_loss`` tracks the best validation loss seen."""
        result, test_ind_score, test_ood_score, representations = self.evaluate_detect(
            dataset_ind, dataset_ood_te, criterion, eval_func, self.device, return_score=True, dataset=dataset, T=T, use_prop=use_prop,
            use_2hop=use_2hop, oodprop=oodprop, oodalpha=oodalpha)
        results[run].append(result)
        if result[-1] < self.min_loss:
            self.min_loss = result[-1]
        if epoch % display_step == 0:
            print(f"Epoch: {epoch:02d}, Loss: {loss:.4f}, AUROC: {100 * result[0]:.2f}%, AUPR: {100 * result[1]:.2f}%, "
                  f"FPR95: {100 * result[2]:.2f}%, Test Score: {100 * result[-2]:.2f}%")
        return result

    def evaluate_detect(self, dataset_ind, dataset_ood, criterion, eval_func, device, return_score, dataset, T, use_prop, use_2hop,
                        oodprop, oodalpha, score_func: Optional[Union[str, Mapping[Any, float]]] = None):
        """scheteronet.py:747-788: negative-energy scores of the in-distribution test nodes against those of the OOD nodes
        (one dataset or a list) -> [auroc, aupr, fpr95] per OOD set, then the test score and the validation loss of the
        classifier.  The encoder passes and the energy propagation run on the device; the measures are host code."""
        from ...base import resolve_score_func
        self.eval()

        def energy(d, idx):
            with torch.no_grad():
                return self.detect(d, idx, device, T, use_prop, use_2hop, oodprop, oodalpha).cpu()

        test_ind_score = energy(dataset_ind, dataset_ind.splits["test"])
        result = []
        for d in (dataset_ood if isinstance(dataset_ood, list) else [dataset_ood]):
            test_ood_score = energy(d, d.node_idx)
            auroc, aupr, fpr, _ = get_measures(test_ind_score, test_ood_score)
            result += [auroc] + [aupr] + [fpr]
        with torch.no_grad():
            out = self(dataset_ind).cpu()
            test_idx = dataset_ind.splits["test"]
            test_score = eval_func(dataset_ind.y[test_idx], out[test_idx], acc=resolve_score_func(score_func or self._DEFAULT_METRIC))
            valid_idx = dataset_ind.splits["valid"]
            if dataset in ("proteins", "ppi"):
                valid_loss = criterion(out[valid_idx], dataset_ind.y[valid_idx].to(torch.float))
            else:
                valid_loss = criterion(F.log_softmax(out[valid_idx], dim=1), dataset_ind.y[valid_idx].squeeze(1))
            result += [test_score] + [valid_loss]
        if return_score:
            return result, test_ind_score, test_ood_score, out.detach().cpu().numpy()
        return result


# ---- data plumbing of the reference's training script (scheteronet.py:155-225, :791-827): host-side bookkeeping, no kernels -----------
class OODData:
    """The fields of the ``torch_geometric.data.Data`` objects the reference hands to ``fit`` / ``detect`` for the out-of-distribution
    nodes (x, edge_index, y, node_idx, num_nodes) — PyG itself is not needed for them."""

    def __init__(self, x, edge_index, y):
        self.x, self.edge_index, self.y = x, edge_index, y
        self.node_idx = None

    @property
    def num_nodes(self):
        return self.x.shape[0]


def get_genename(raw_adata):
    """:791-798."""
    for key in ("gene_id", "symbol"):
        if key in raw_adata.var.keys():
            return raw_adata.var[key].values
    return raw_adata.var.index


def set_split(data, train_idx=[], val_idx=[], test_idx=[]):
    """:801-827: the rarest cell type becomes the out-of-distribution class; its cells leave the train / val / test index lists, which
    are stored in ``uns`` next to ``ood_idx`` / ``id_idx``; labels, gene names and per-cell counts are written to ``obs`` / ``var``."""
    from collections import Counter

    import pandas as pd
    adata = data.data
    adata.obs["assay"] = "10x 3' v2"
    ct = adata.obsm["cell_type"]
    y = np.argmax(ct.values if isinstance(ct, pd.DataFrame) else np.asarray(ct), axis=1)
    adata.obs["cell_type_raw"] = y
    for key in list(adata.obsm.keys()):
        if isinstance(adata.obsm[key], pd.DataFrame):
            adata.obsm[key] = adata.obsm[key].values
    adata.obs["cell"] = y
    adata.var["gene_name"] = get_genename(adata)
    ood_class = min(Counter(y.tolist()).items(), key=lambda kv: kv[1])[0]
    ood = set(np.flatnonzero(y == ood_class).tolist())
    adata.uns["train_idx"] = [int(i) for i in train_idx if i not in ood]
    adata.uns["val_idx"] = [int(i) for i in val_idx if i not in ood]
    adata.uns["test_idx"] = [int(i) for i in test_idx if i not in ood]
    adata.uns["ood_idx"] = sorted(ood)
    adata.uns["id_idx"] = [int(i) for i in np.flatnonzero(y != ood_class)]
    x = adata.X
    adata.obs["n_counts"] = np.asarray(x.sum(axis=1)).ravel() if not hasattr(x, "tensor") else x.tensor.sum(1).cpu().numpy()


def convert_dgl_to_original_format(g, adata, ref_adata_name: str):
    """:155-225 for the graph ``HeteronetGraph`` leaves in ``uns`` (a CellGeneGraph with ``feat`` / ``label`` node data instead of a
    DGLGraph): the in-distribution ``NCDataset`` and the two out-of-distribution views over the same nodes and edges."""
    features, labels = g.ndata["feat"], g.ndata["label"]
    src, dst = g.edges()
    edge_index = torch.stack([src, dst], dim=0)
    dataset_ind = NCDataset(ref_adata_name)
    dataset_ind.graph = {"edge_index": edge_index, "edge_feat": None, "node_feat": features, "num_nodes": g.number_of_nodes()}
    dataset_ind.label = labels
    dataset_ind.num_nodes = g.number_of_nodes()
    dataset_ind.edge_index, dataset_ind.x, dataset_ind.y = edge_index, features, labels
    dataset_ind.batch = g.ndata["batch_id"] if "batch_id" in g.ndata else None
    dataset_ind.node_idx = adata.uns["id_idx"]
    dataset_ind.splits = {"train": adata.uns["train_idx"], "valid": adata.uns["val_idx"], "test": adata.uns["test_idx"]}
    dataset_ood_tr, dataset_ood_te = OODData(features, edge_index, labels), OODData(features, edge_index, labels)
    dataset_ood_tr.node_idx = dataset_ood_te.node_idx = adata.uns["ood_idx"]
    logger.info(f"NCDataset (ID): Nodes={dataset_ind.num_nodes}, Edges={edge_index.shape[1]}; OOD nodes referenced={len(adata.uns['ood_idx'])}")
    return dataset_ind, dataset_ood_tr, dataset_ood_te, adata


def stable_cumsum(arr, rtol=1e-05, atol=1e-08):
    """float64 cumulative sum whose last element is checked against the plain sum (scheteronet.py:1034-1052)."""
    out = np.cumsum(arr, dtype=np.float64)
    if not np.allclose(out[-1], np.sum(arr, dtype=np.float64), rtol=rtol, atol=atol):
        raise RuntimeError("cumsum was found to be unstable: its last element does not correspond to sum")
    return out


def fpr_and_fdr_at_recall(y_true, y_score, recall_level=0.95, pos_label=None):
    """False-positive rate (and the score threshold) at the operating point whose recall is closest to ``recall_level``
    (scheteronet.py:1055-1094).  Thresholds sit at the last sample of every run of tied scores, walked from the first one
    reaching full recall towards the highest score, with a (recall 1, no false positives) sentinel at the end — the order
    matters for argmin's tie-breaking, so it is kept."""
    classes = np.unique(y_true)
    binary = any(np.array_equal(classes, c) for c in ([0, 1], [-1, 1], [0], [-1], [1]))
    if pos_label is None and not binary:
        raise ValueError("Data is not binary and pos_label is not specified")
    if pos_label is None:
        pos_label = 1.
    positive = (y_true == pos_label)
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score, positive = y_score[order], positive[order]
    run_ends = np.r_[np.where(np.diff(y_score))[0], positive.size - 1]
    tps = stable_cumsum(positive)[run_ends]
    fps = 1 + run_ends - tps
    thresholds = y_score[run_ends]
    recall = tps / tps[-1]
    back = slice(tps.searchsorted(tps[-1]), None, -1)
    recall, fps, thresholds = np.r_[recall[back], 1], np.r_[fps[back], 0], thresholds[back]
    cutoff = np.argmin(np.abs(recall - recall_level))
    if np.array_equal(classes, [1]):
        return thresholds[cutoff]
    return fps[cutoff] / (np.sum(np.logical_not(positive))), thresholds[cutoff]


def get_measures(_pos, _neg, recall_level=0.95):
    """(auroc, aupr, fpr@recall, threshold) of "positive = in-distribution" from two score vectors (scheteronet.py:1097-1108)."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    pos = np.array(_pos[:]).reshape((-1, 1))
    neg = np.array(_neg[:]).reshape((-1, 1))
    examples = np.squeeze(np.vstack((pos, neg)))
    labels = np.zeros(len(examples), dtype=np.int32)
    labels[:len(pos)] += 1
    fpr, threshold = fpr_and_fdr_at_recall(labels, examples, recall_level)
    return roc_auc_score(labels, examples), average_precision_score(labels, examples), fpr, threshold
