"""The graph layers of scGNN 2.0 on MI355X — drop-in for the GNN part of dance/modules/single_modality/imputation/scgnn2.py:
``GraphConvolution`` (:479-505, the GCN-(V)AE layer: dropout -> mm -> spmm -> act), ``InnerProductDecoder`` (:418-429),
``GCNModelVAE`` / ``GCNModelAE`` (:438-478), the hand-rolled multi-head graph attention ``GATLayer`` / ``GAT`` (:883-1189) and
``Graph_AE`` (:373-415) that combines them.  Same constructors, parameter names (``weight``; ``linear_proj.weight``,
``scoring_fn_target``, ``scoring_fn_source``, ``bias``, ``skip_proj.weight``) and ``forward`` conventions (``GATLayer`` takes and
returns the ``(features, edge_index)`` tuple so that ``nn.Sequential`` can chain it).

What runs where: ``GraphConvolution`` is the fused GCN layer op (dh_gemm_f32 + dh_spmm_csr_f32 with the ReLU in the SpMM epilogue,
hand-written backward, autograd.gcn_layer).  ``GATLayer`` never builds the reference's [E, NH, FOUT] lifted / weighted feature
tensors: per head it is dh_edge_softmax_shift_f32 (leaky-ReLU scores, exponent shifted by the GLOBAL maximum score and a
+ 1e-16 denominator exactly as :1071-1085 — not the usual row-shifted softmax) + the CSR SpMM with the attention as edge values
(autograd.gat_aggregate; backward through dh_sddmm_csr_f32 and dh_edge_softmax_backward_f32); projections are HipLinear GEMMs.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from .... import kernels
from ....autograd import HipLinear, gat_aggregate, gcn_layer, linear
from ....graph import CSRGraph, as_graph
from ...spatial.spatial_domain.stagate import edge_index_graph


def _is_relu(act):
    return act in (F.relu, torch.relu) or isinstance(act, nn.ReLU)


class GraphConvolution(nn.Module):
    """Simple GCN layer, similar to https://arxiv.org/abs/1609.02907 (scgnn2.py:479-505)."""

    def __init__(self, in_features, out_features, dropout=0., act=F.relu):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.dropout = dropout
        self.act = act
        self.weight = Parameter(torch.empty(in_features, out_features))
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.weight)

    def forward(self, input, adj):
        input = F.dropout(input, self.dropout, self.training)
        graph = adj if isinstance(adj, CSRGraph) else as_graph(adj, input.device)
        relu = _is_relu(self.act)
        output = gcn_layer(input, self.weight, graph, None, relu)
        return output if relu else self.act(output)

    def __repr__(self):
        return self.__class__.__name__ + " (" + str(self.in_features) + " -> " + str(self.out_features) + ")"


class InnerProductDecoder(nn.Module):
    """Decoder for using inner product for prediction (scgnn2.py:418-429): act(z z^T) on the matrix cores."""

    def __init__(self, dropout, act=torch.sigmoid):
        super().__init__()
        self.dropout = dropout
        self.act = act

    def forward(self, z):
        z = F.dropout(z, self.dropout, training=self.training)
        return self.act(linear(z, z))


class GCNModelVAE(nn.Module):

    def __init__(self, input_feat_dim, hidden_dim1, hidden_dim2, dropout):
        super().__init__()
        self.gc1 = GraphConvolution(input_feat_dim, hidden_dim1, dropout, act=F.relu)
        self.gc2 = GraphConvolution(hidden_dim1, hidden_dim2, dropout, act=lambda x: x)
        self.gc3 = GraphConvolution(hidden_dim1, hidden_dim2, dropout, act=lambda x: x)
        self.dc = InnerProductDecoder(dropout, act=lambda x: x)

    def encode(self, x, adj):
        hidden1 = self.gc1(x, adj)
        return self.gc2(hidden1, adj), self.gc3(hidden1, adj)

    def reparameterize(self, mu, logvar):
        if self.training:
            std = torch.exp(logvar)
            eps = torch.randn_like(std)
            return eps.mul(std).add_(mu)
        return mu

    def forward(self, x, adj):
        mu, logvar = self.encode(x, adj)
        z = self.reparameterize(mu, logvar)
        return z, mu, logvar


class GCNModelAE(nn.Module):

    def __init__(self, input_feat_dim, hidden_dim1, hidden_dim2, dropout):
        super().__init__()
        self.gc1 = GraphConvolution(input_feat_dim, hidden_dim1, dropout, act=F.relu)
        self.gc2 = GraphConvolution(hidden_dim1, hidden_dim2, dropout, act=lambda x: x)
        self.dc = InnerProductDecoder(dropout, act=lambda x: x)

    def encode(self, x, adj):
        hidden1 = self.gc1(x, adj)
        return self.gc2(hidden1, adj)

    def forward(self, x, adj, encode=False):
        z = self.encode(x, adj)
        return z, z, None


class GATLayer(nn.Module):
    """Multi-head graph attention layer (scgnn2.py:920-1189, "implementation #3")."""

    src_nodes_dim = 0  # position of source nodes in edge index
    trg_nodes_dim = 1  # position of target nodes in edge index
    nodes_dim = 0
    head_dim = 1

    def __init__(self, num_in_features, num_out_features, num_of_heads, concat=True, activation=nn.ELU(), dropout_prob=0.6,
                 add_skip_connection=True, bias=True, log_attention_weights=False):
        super().__init__()
        self.num_of_heads = num_of_heads
        self.num_out_features = num_out_features
        self.concat = concat
        self.add_skip_connection = add_skip_connection
        self.linear_proj = HipLinear(num_in_features, num_of_heads * num_out_features, bias=False)
        self.scoring_fn_target = nn.Parameter(torch.empty(1, num_of_heads, num_out_features))
        self.scoring_fn_source = nn.Parameter(torch.empty(1, num_of_heads, num_out_features))
        if bias and concat:
            self.bias = nn.Parameter(torch.empty(num_of_heads * num_out_features))
        elif bias and not concat:
            self.bias = nn.Parameter(torch.empty(num_out_features))
        else:
            self.register_parameter("bias", None)
        if add_skip_connection:
            self.skip_proj = HipLinear(num_in_features, num_of_heads * num_out_features, bias=False)
        else:
            self.register_parameter("skip_proj", None)
        self.leakyReLU = nn.LeakyReLU(0.2)
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_prob)
        self.log_attention_weights = log_attention_weights
        self.attention_weights = None
        self.init_params()

    def init_params(self):
        nn.init.xavier_uniform_(self.linear_proj.weight)
        nn.init.xavier_uniform_(self.scoring_fn_target)
        nn.init.xavier_uniform_(self.scoring_fn_source)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, data):
        in_nodes_features, edge_index = data
        num_of_nodes = in_nodes_features.shape[self.nodes_dim]
        assert edge_index.shape[0] == 2, f"Expected edge index with shape=(2,E) got {edge_index.shape}"
        nh, fo = self.num_of_heads, self.num_out_features
        in_nodes_features = self.dropout(in_nodes_features)
        proj = self.dropout(self.linear_proj(in_nodes_features))  # [N, NH * FOUT]; head h = columns h * FOUT ...
        proj3 = proj.view(-1, nh, fo)
        scores_source = (proj3 * self.scoring_fn_source).sum(dim=-1)  # [N, NH]
        scores_target = (proj3 * self.scoring_fn_target).sum(dim=-1)
        graph, slot = edge_index_graph(edge_index, num_of_nodes)  # CSR by target node, duplicates kept; cached per edge_index
        # the reference subtracts the maximum over ALL edges and heads before exp (:1072): one scalar, computed here from the
        # [E, NH] scores (the only per-edge tensor this layer materialises besides the attentions themselves)
        with torch.no_grad():
            src, trg = edge_index[self.src_nodes_dim].long(), edge_index[self.trg_nodes_dim].long()
            shift = self.leakyReLU(scores_source.index_select(0, src) + scores_target.index_select(0, trg)).max().reshape(1)
        heads, atts = [], []
        for h in range(nh):
            keep = None
            if self.training and self.dropout.p > 0:  # dropout on the attention coefficients (:1019), CSR edge order
                keep = self.dropout(torch.ones(graph.nnz, dtype=torch.float32, device=proj.device))
            out_h, att_h = gat_aggregate(proj[:, h * fo:(h + 1) * fo], scores_source[:, h], scores_target[:, h], graph,
                                         act=kernels.ATT_LEAKY_RELU, negative_slope=0.2, shift=shift, edge_scale=keep)
            heads.append(out_h)
            atts.append(att_h if keep is None else att_h * keep)
        out_nodes_features = torch.stack(heads, dim=1)  # [N, NH, FOUT]
        # [E, NH, 1] in the ORIGINAL edge order, as the reference hands them on (only gathered when they are logged)
        attentions_per_edge = torch.stack(atts, dim=1)[slot].unsqueeze(-1) if self.log_attention_weights else None
        out_nodes_features = self.skip_concat_bias(attentions_per_edge, in_nodes_features, out_nodes_features)
        return (out_nodes_features, edge_index)

    def skip_concat_bias(self, attention_coefficients, in_nodes_features, out_nodes_features):
        if self.log_attention_weights:
            self.attention_weights = attention_coefficients
        if self.add_skip_connection:
            if out_nodes_features.shape[-1] == in_nodes_features.shape[-1]:
                out_nodes_features = out_nodes_features + in_nodes_features.unsqueeze(1)
            else:
                out_nodes_features = out_nodes_features + self.skip_proj(in_nodes_features).view(-1, self.num_of_heads, self.num_out_features)
        if self.concat:
            out_nodes_features = out_nodes_features.reshape(-1, self.num_of_heads * self.num_out_features)
        else:
            out_nodes_features = out_nodes_features.mean(dim=self.head_dim)
        if self.bias is not None:
            out_nodes_features = out_nodes_features + self.bias
        return out_nodes_features if self.activation is None else self.activation(out_nodes_features)


class GAT(torch.nn.Module):
    """Stack of GATLayers (scgnn2.py:883-917): concat + ELU between layers, head average and raw scores at the end."""

    def __init__(self, num_of_layers, num_heads_per_layer, num_features_per_layer, add_skip_connection=True, bias=True, dropout=0.6,
                 log_attention_weights=False):
        super().__init__()
        assert num_of_layers == len(num_heads_per_layer) == len(num_features_per_layer) - 1, "Enter valid arch params."
        num_heads_per_layer = [1] + num_heads_per_layer
        gat_layers = []
        for i in range(num_of_layers):
            gat_layers.append(
                GATLayer(num_in_features=num_features_per_layer[i] * num_heads_per_layer[i], num_out_features=num_features_per_layer[i + 1],
                         num_of_heads=num_heads_per_layer[i + 1], concat=True if i < num_of_layers - 1 else False,
                         activation=nn.ELU() if i < num_of_layers - 1 else None, dropout_prob=dropout,
                         add_skip_connection=add_skip_connection, bias=bias, log_attention_weights=log_attention_weights))
        self.gat_net = nn.Sequential(*gat_layers)

    def forward(self, data):
        return self.gat_net(data)


class Graph_AE(nn.Module):
    """scgnn2.py:373-415: GAT encoder (default) or GCN-VAE encoder + inner-product graph decoder."""

    def __init__(self, dim, embedding_size, gat_dropout=0, multi_heads=2, gat_hid_embed=64):
        super().__init__()
        self.gat = GAT(num_of_layers=2, num_heads_per_layer=[multi_heads, multi_heads],
                       num_features_per_layer=[dim, gat_hid_embed, embedding_size], dropout=gat_dropout)
        self.gc1 = GraphConvolution(dim, 32, 0, act=F.relu)
        self.gc2 = GraphConvolution(32, embedding_size, 0, act=lambda x: x)
        self.gc3 = GraphConvolution(32, embedding_size, 0, act=lambda x: x)
        self.decode = InnerProductDecoder(0, act=lambda x: x)

    def encode_gat(self, in_nodes_features, edge_index):
        return self.gat((in_nodes_features, edge_index))

    def encode_gae(self, x, adj):
        hidden1 = self.gc1(x, adj)
        return self.gc2(hidden1, adj), self.gc3(hidden1, adj)

    def reparameterize(self, mu, logvar):
        if self.training:
            std = torch.exp(logvar)
            eps = torch.randn_like(std)
            return eps.mul(std).add_(mu)
        return mu

    def forward(self, in_nodes_features, edge_index, encode=False, use_GAT=True):
        gae_info = None
        if use_GAT:
            out_nodes_features = self.encode_gat(in_nodes_features, edge_index)[0]
        else:
            gae_info = self.encode_gae(in_nodes_features, edge_index)
            out_nodes_features = self.reparameterize(*gae_info)
        recon_graph = self.decode(out_nodes_features)
        return out_nodes_features, gae_info, recon_graph
