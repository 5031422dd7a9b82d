"""CPU restatements (plain torch, fp32) of the model-level paths bench.py times as ``cpu_baseline`` for BASELINE configs 2, 3, 5.

TEST INFRASTRUCTURE ONLY: imported by tests/ (pinned against the goldens the reference's own classes produced:
tests/golden/model_heads.npz, gc_dec.npz, scdeepsort.npz — tests/test_oracle_models.py) and by bench.py's ``cpu_baseline``
legs (``kind: "port"``; the AST-lifted reference classes cannot travel to the GPU box, these can).  Nothing under dance_amd/
imports this module.  Every block cites the reference lines it follows (paths relative to /root/reference).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import GNNLayer, GraphConvolution


# ---- scDSC (dance/modules/single_modality/clustering/scdsc.py) ------------------------------------------------------------------
class AE(nn.Module):
    """scdsc.py:504-598: 3 encoder + 3 bottleneck + 3 decoder Linear+BatchNorm1d blocks and the reconstruction layer."""

    def __init__(self, n_enc_1, n_enc_2, n_enc_3, n_dec_1, n_dec_2, n_dec_3, n_input, n_z1, n_z2, n_z3):
        super().__init__()
        dims = [("enc_1", n_input, n_enc_1), ("enc_2", n_enc_1, n_enc_2), ("enc_3", n_enc_2, n_enc_3), ("z1_layer", n_enc_3, n_z1),
                ("z2_layer", n_z1, n_z2), ("z3_layer", n_z2, n_z3), ("dec_1", n_z3, n_dec_1), ("dec_2", n_dec_1, n_dec_2), ("dec_3", n_dec_2, n_dec_3)]
        for i, (name, a, b) in enumerate(dims, 1):   # :535-554, same attribute names so reference checkpoints load
            setattr(self, name, nn.Linear(a, b))
            setattr(self, f"BN{i}", nn.BatchNorm1d(b))
        self.x_bar_layer = nn.Linear(n_dec_3, n_input)

    def forward(self, x):   # :584-598
        enc_h1 = F.relu(self.BN1(self.enc_1(x)))
        enc_h2 = F.relu(self.BN2(self.enc_2(enc_h1)))
        enc_h3 = F.relu(self.BN3(self.enc_3(enc_h2)))
        z1 = self.BN4(self.z1_layer(enc_h3))
        z2 = self.BN5(self.z2_layer(z1))
        z3 = self.BN6(self.z3_layer(z2))
        dec_h1 = F.relu(self.BN7(self.dec_1(z3)))
        dec_h2 = F.relu(self.BN8(self.dec_2(dec_h1)))
        dec_h3 = F.relu(self.BN9(self.dec_3(dec_h2)))
        return self.x_bar_layer(dec_h3), enc_h1, enc_h2, enc_h3, z3, z2, z1, dec_h3


class _MeanAct(nn.Module):   # scdsc.py:601-608
    def forward(self, x):
        return torch.clamp(torch.exp(x), min=1e-5, max=1e6)


class _DispAct(nn.Module):   # scdsc.py:611-618
    def forward(self, x):
        return torch.clamp(F.softplus(x), min=1e-4, max=1e4)


def zinb_loss(x, mean, disp, pi, scale_factor, ridge_lambda=0.0):
    """dance/utils/loss.py:780-829 (the lgamma terms in float64, as written there)."""
    eps = 1e-10
    mean = mean * scale_factor[:, None]
    t1 = torch.lgamma(disp.double() + eps) + torch.lgamma(x.double() + 1.0) - torch.lgamma(x.double() + disp.double() + eps)
    t2 = (disp + x) * torch.log(1.0 + (mean / (disp + eps))) + (x * (torch.log(disp + eps) - torch.log(mean + eps)))
    nb_case = t1 + t2 - torch.log(1.0 - pi + eps)
    zero_nb = torch.pow(disp / (disp + mean + eps), disp)
    zero_case = -torch.log(pi + ((1.0 - pi) * zero_nb) + eps)
    result = torch.where(torch.le(x, 1e-8), zero_case, nb_case)
    if ridge_lambda > 0:
        result = result + ridge_lambda * torch.square(pi)
    return torch.mean(result)


class ScDSCModel(nn.Module):
    """scdsc.py:339-472: autoencoder + 7 chained GNNLayers mixed with its activations + ZINB heads + Student-t assignment."""

    def __init__(self, sigma=1, n_enc_1=512, n_enc_2=256, n_enc_3=256, n_dec_1=256, n_dec_2=256, n_dec_3=512, n_z1=256, n_z2=128, n_z3=32,
                 n_clusters=10, n_input=100, v=1):
        super().__init__()
        self.sigma, self.v = sigma, v
        self.ae = AE(n_enc_1, n_enc_2, n_enc_3, n_dec_1, n_dec_2, n_dec_3, n_input, n_z1, n_z2, n_z3)
        widths = [n_input, n_enc_1, n_enc_2, n_enc_3, n_z1, n_z2, n_z3, n_clusters]   # :403-409
        for i in range(7):
            setattr(self, f"gnn_{i + 1}", GNNLayer(widths[i], widths[i + 1]))
        self.cluster_layer = nn.Parameter(torch.empty(n_clusters, n_z3))   # :412-413
        nn.init.xavier_normal_(self.cluster_layer.data)
        self._dec_mean = nn.Sequential(nn.Linear(n_dec_3, n_input), _MeanAct())   # :414-416
        self._dec_disp = nn.Sequential(nn.Linear(n_dec_3, n_input), _DispAct())
        self._dec_pi = nn.Sequential(nn.Linear(n_dec_3, n_input), nn.Sigmoid())

    def forward(self, x, adj):   # :446-472
        x_bar, tra1, tra2, tra3, z3, z2, z1, dec_h3 = self.ae(x)
        s = self.sigma
        h = self.gnn_1(x, adj)
        h = self.gnn_2((1 - s) * h + s * tra1, adj)
        h = self.gnn_3((1 - s) * h + s * tra2, adj)
        h = self.gnn_4((1 - s) * h + s * tra3, adj)
        h = self.gnn_5((1 - s) * h + s * z1, adj)
        h = self.gnn_6((1 - s) * h + s * z2, adj)
        h = self.gnn_7((1 - s) * h + s * z3, adj, active=False)
        predict = F.softmax(h, dim=1)
        _mean, _disp, _pi = self._dec_mean(dec_h3), self._dec_disp(dec_h3), self._dec_pi(dec_h3)
        q = 1.0 / (1.0 + torch.sum(torch.pow(z3.unsqueeze(1) - self.cluster_layer, 2), 2) / self.v)
        q = q.pow((self.v + 1.0) / 2.0)
        q = (q.t() / torch.sum(q, 1)).t()
        return x_bar, q, predict, z3, _mean, _disp, _pi


def scdsc_target(q):   # scdsc.py:196-198
    p = q**2 / q.sum(0)
    return (p.t() / p.sum(1)).t()


def scdsc_epoch(model, optimizer, data, adj, x_raw, sf, p, bcl=0.1, cl=0.01, rl=1.0, zl=0.1):
    """One iteration of the joint loop, scdsc.py:270-287: forward, the four-term loss, backward, Adam step."""
    model.train()
    x_bar, q, pred, _, mean, disp, pi = model(data, adj)
    loss = (bcl * F.binary_cross_entropy(q, p) + cl * F.kl_div(pred.log(), p, reduction="batchmean") + rl * F.mse_loss(x_bar, data)
            + zl * zinb_loss(x_raw, mean, disp, pi, sf))
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(loss.detach())


# ---- SpaGCN (dance/modules/spatial/spatial_domain/spagcn.py) ---------------------------------------------------------------------
class SimpleGCDEC(nn.Module):
    """spagcn.py:369-425: one GraphConvolution + the DEC head, with the reference's literal precedence in ``q`` (:394-396)."""

    def __init__(self, nfeat, nhid, alpha=0.2):
        super().__init__()
        self.gc = GraphConvolution(nfeat, nhid)
        self.alpha = alpha
        self.mu = nn.Parameter(torch.zeros(1, nhid))   # set by the initial clustering in fit (:471-500)

    def forward(self, x, adj):
        x = self.gc(x, adj)
        q = 1.0 / ((1.0 + torch.sum((x.unsqueeze(1) - self.mu)**2, dim=2) / self.alpha) + 1e-8)
        q = q**(self.alpha + 1.0) / 2.0
        return x, q / torch.sum(q, dim=1, keepdim=True)

    @staticmethod
    def loss_function(p, q):   # :398-406
        return torch.mean(torch.sum(p * torch.log(p / (q + 1e-6)), dim=1))

    @staticmethod
    def target_distribution(q):   # :408-425
        p = q**2 / torch.sum(q, dim=0)
        return p / torch.sum(p, dim=1, keepdim=True)


def spagcn_iteration(model, optimizer, x, adj, p):
    """One epoch of SimpleGCDEC.fit's loop, spagcn.py:518-530: forward, KL loss, backward, step, label read-back."""
    optimizer.zero_grad()
    _, q = model(x, adj)
    loss = model.loss_function(p, q)
    loss.backward()
    optimizer.step()
    return torch.argmax(q, dim=1).detach().cpu().numpy(), float(loss.detach())


# ---- scDeepSort (dance/modules/single_modality/cell_type_annotation/scdeepsort.py, dance/models/nn/gnn.py) ---------------------
class ScDeepSortGNN(nn.Module):
    """One-layer GNN of scdeepsort.py:26-88 with AdaptiveSAGE (gnn.py:8-96): the weighted mean ``neigh`` is computed and — as
    the reference writes it (:92) — NOT used; the output is Linear(relu(Linear(h_dst)))."""

    def __init__(self, dim_in, dim_hid, dim_out, gene_num):
        super().__init__()
        self.gene_num = gene_num
        self.alpha = nn.Parameter(torch.ones(gene_num + 2, 1))
        self.lin = nn.Linear(dim_in, dim_hid)
        nn.init.xavier_uniform_(self.lin.weight, gain=nn.init.calculate_gain("relu"))
        self.linear = nn.Linear(dim_hid, dim_out)
        nn.init.xavier_uniform_(self.linear.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, h_src, n_dst, e_src, e_dst, w, src_cell_id):
        # message_func (gnn.py:62-82) on a block whose destinations are the first n_dst source nodes
        sid, did = src_cell_id[e_src], src_cell_id[e_dst]
        idx = torch.full_like(sid, self.gene_num + 1, dtype=torch.long)            # cell self loop
        idx = torch.where((sid >= 0) & (did < 0), sid.long(), idx)                 # gene -> cell
        idx = torch.where((did >= 0) & (sid < 0), did.long(), idx)                 # cell -> gene
        idx = torch.where((did >= 0) & (sid >= 0), torch.full_like(idx, self.gene_num), idx)   # gene self loop
        m = h_src[e_src] * self.alpha[idx] * w[:, None]
        neigh = torch.zeros((n_dst, h_src.shape[1]), dtype=h_src.dtype).index_add_(0, e_dst, m)
        neigh = neigh / torch.bincount(e_dst, minlength=n_dst).clamp(min=1)[:, None]   # fn.mean (gnn.py:90)
        self.last_neigh = neigh
        return self.linear(F.relu(self.lin(h_src[:n_dst])))                       # gnn.py:92-96, scdeepsort.py:84-88


def scdeepsort_block(rowptr, col, val, seeds):
    """Full fan-out in-neighbour block of ``seeds`` (dgl NeighborSampler([-1]), scdeepsort.py:183): destinations first among the
    sources, then the other in-neighbours (ascending).  Returns (src_ids, e_src, e_dst, w) as numpy, edges in CSR order."""
    seeds = np.asarray(seeds, dtype=np.int64)
    starts, ends = rowptr[seeds], rowptr[seeds + 1]
    cnt = ends - starts
    e = np.concatenate([np.arange(a, b) for a, b in zip(starts, ends)]) if seeds.size else np.zeros(0, dtype=np.int64)
    dst = np.repeat(np.arange(seeds.size), cnt)
    src_nodes = col[e].astype(np.int64)
    lut = {int(s): i for i, s in enumerate(seeds)}
    extra = np.setdiff1d(np.unique(src_nodes), seeds)
    for i, s in enumerate(extra):
        lut[int(s)] = seeds.size + i
    e_src = np.fromiter((lut[int(s)] for s in src_nodes), dtype=np.int64, count=src_nodes.size)
    return np.concatenate((seeds, extra)), e_src, dst, val[e]


def scdeepsort_batch(model, optimizer, rowptr, col, val, features, cell_id, labels, seeds):
    """One training batch of ScDeepSort.cal_loss (scdeepsort.py:238-246): block, forward, summed CE, Adam step."""
    src_ids, e_src, e_dst, w = scdeepsort_block(rowptr, col, val, seeds)
    logits = model(features[src_ids], len(seeds), torch.from_numpy(e_src), torch.from_numpy(e_dst), torch.from_numpy(w.astype(np.float32)),
                   cell_id[src_ids])
    loss = F.cross_entropy(logits, labels[seeds], reduction="sum")
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(loss.detach())


# ---- graph-sc (dance/modules/single_modality/clustering/graphsc.py) ------------------------------------------------------------------
class GraphSCAE(nn.Module):
    """``GCNAE`` of graphsc.py:274-384 with one ``WeightedGraphConv`` layer (:428-484, dgl GraphConv norm="both", bias, relu), the hidden
    Linear stack and the identity-activation ``InnerProductDecoder`` (:387-425; its dropout is a constructor argument here, 0 where a
    comparison must be deterministic).  State-dict names follow the reference (``layer1.weight / bias``, ``encoder.<i>.weight / bias``)."""

    def __init__(self, in_feats, hidden_dim=200, hidden=(300, ), agg="sum", decoder_dropout=0.1, dropout=0.0):
        super().__init__()
        self.agg, self.decoder_dropout, self.dropout = agg, decoder_dropout, dropout
        self.layer1 = nn.Module()
        self.layer1.weight = nn.Parameter(torch.empty(in_feats, hidden_dim))
        self.layer1.bias = nn.Parameter(torch.zeros(hidden_dim))
        nn.init.xavier_uniform_(self.layer1.weight)
        dims = (hidden_dim, ) + tuple(hidden)
        self.encoder = nn.Sequential(*[nn.Linear(dims[i], dims[i + 1]) for i in range(len(hidden))]) if hidden else None

    def forward(self, h_src, n_dst, e_src, e_dst, w):
        n_src = h_src.shape[0]
        if self.dropout and self.training:
            h_src = F.dropout(h_src, self.dropout)                                     # GCNAE.dropout on the block's input (:366-367)
        out_deg = torch.bincount(e_src, minlength=n_src).float().clamp(min=1)        # degrees INSIDE the block, as DGL computes them
        in_deg = torch.bincount(e_dst, minlength=n_dst).float().clamp(min=1)
        feat = (h_src * out_deg.pow(-0.5)[:, None]) @ self.layer1.weight              # :452-467
        m = feat[e_src] * w[:, None]                                                   # edge_selection_simple (:430-438)
        h = torch.zeros((n_dst, feat.shape[1]), dtype=feat.dtype).index_add_(0, e_dst, m)
        if self.agg == "mean":
            h = h / torch.bincount(e_dst, minlength=n_dst).clamp(min=1)[:, None]
        h = F.relu(h * in_deg.pow(-0.5)[:, None] + self.layer1.bias)                   # :472-483
        z = self.encoder(h) if self.encoder is not None else h
        zd = F.dropout(z, self.decoder_dropout, training=True) if self.decoder_dropout else z   # F.dropout(z, p): always on (:423)
        return zd @ zd.t(), z


def graphsc_batch(model, optimizer, rowptr, col, val, features, seeds):
    """One batch of ``GraphSC.fit`` (graphsc.py:196-220): full-neighbour block of the seed cells, a first forward (its embedding is what
    the loop collects), the dense dst x dst adjacency of the block (= the seeds' self loops in a cell-gene graph), the weighted BCE on the
    logits of a SECOND forward, Adam step.  Returns (loss, embedding of the first forward)."""
    src_ids, e_src, e_dst, w = scdeepsort_block(rowptr, col, val, seeds)
    es, ed, wt = torch.from_numpy(e_src), torch.from_numpy(e_dst), torch.from_numpy(w.astype(np.float32))
    n_dst = len(seeds)
    with torch.no_grad():
        _, emb = model(features[src_ids], n_dst, es, ed, wt)
    adj = torch.zeros((n_dst, n_dst))
    inner = e_src < n_dst
    adj[ed[inner], es[inner]] = 1.0
    tot = float(adj.sum())
    pos_weight = torch.tensor([(n_dst * n_dst - tot) / tot])
    factor = (n_dst * n_dst - tot) * 2 or 1
    logits, _ = model(features[src_ids], n_dst, es, ed, wt)
    loss = (n_dst * n_dst / factor) * F.binary_cross_entropy_with_logits(logits, adj, pos_weight=pos_weight)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(loss.detach()), emb
