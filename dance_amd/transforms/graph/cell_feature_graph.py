"""CellFeatureGraph / PCACellFeatureGraph on MI355X — drop-ins for
dance/transforms/graph/cell_feature_graph.py:12-112.

The reference builds a DGL graph and rescales in-edge weights with a Python loop over every node (:62-68).  Here
the expression matrix goes to the device as CSR once; its transpose (dh_csr_transpose), the per-node rescale
(dh_csr_row_normalize_f32, two launches) and the assembly with self loops in the reference's edge order
(dh_cellgene_graph_assemble) are HIP kernels.  The result is a ``CellGeneGraph`` (DGLGraph stand-in) in
``data.data.uns[out]`` carrying ``ndata["cell_id" | "feat_id" | "features"]`` and edge weights.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch

from ... import kernels
from ...cellgraph import CellGeneGraph
from ...registry import register_preprocessor
from ..base import BaseTransform
from ..cell_feature import WeightedFeaturePCA


@register_preprocessor("graph", "cell")
class CellFeatureGraph(BaseTransform):

    def __init__(self, cell_feature_channel: str, gene_feature_channel=None, *, mod=None, normalize_edges: bool = True,
                 device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.cell_feature_channel = cell_feature_channel
        self.gene_feature_channel = gene_feature_channel or cell_feature_channel
        self.mod = mod
        self.normalize_edges = normalize_edges
        self.device = device

    def __call__(self, data):
        from ...data import DeviceArray
        feat = data.get_feature(return_type="default", channel_type="X", mod=self.mod)
        dev = self.device
        if isinstance(feat, DeviceArray):
            # the expression matrix is already on the device (on-device preprocessing): its non-zeros in np.nonzero order (:38)
            # come from dh_dense_nnz_count_f32 / dh_dense_to_csr_f32 — no host copy of X
            xd = feat.tensor.to(device=dev, dtype=torch.float32)
            num_cells, num_feats = xd.shape
            rp_x, col_x, val_x = kernels.dense_to_csr(xd)
            nnz = int(col_x.numel())
        else:
            x = sp.csr_matrix(feat, dtype=np.float32)
            x.eliminate_zeros()  # np.nonzero semantics (:38)
            x.sort_indices()  # row-major nonzero order
            num_cells, num_feats = x.shape
            nnz = x.nnz
            rp_x = torch.from_numpy(x.indptr.astype(np.int32)).to(dev)
            col_x = torch.from_numpy(x.indices.astype(np.int32)).to(dev)
            val_x = torch.from_numpy(x.data.astype(np.float32)).to(dev)
        self.logger.info(f"Number of nonzero entries: {nnz:,}")
        self.logger.info(f"Nonzero rate = {nnz / num_cells / num_feats:.1%}")
        rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col_x, val_x, num_cells, num_feats)
        if self.normalize_edges:  # in-degree rescale before the self loops are added (:62-68)
            val_x = kernels.csr_row_normalize(rp_x, val_x)  # cells' in-edges (gene -> cell)
            val_t = kernels.csr_row_normalize(rp_t, val_t)  # genes' in-edges (cell -> gene)
        rowptr, col, val, eid = kernels.cellgene_graph_assemble(rp_x, col_x, val_x, rp_t, col_t, val_t, perm_t,
                                                                num_cells, num_feats)
        cell_id = torch.cat((torch.arange(num_feats, dtype=torch.int32), -torch.ones(num_cells, dtype=torch.int32)))
        feat_id = torch.cat((-torch.ones(num_feats, dtype=torch.int32), torch.arange(num_cells, dtype=torch.int32)))
        gene_feature = data.get_feature(return_type=dev, channel=self.gene_feature_channel, mod=self.mod, channel_type="varm")
        cell_feature = data.get_feature(return_type=dev, channel=self.cell_feature_channel, mod=self.mod, channel_type="obsm")
        features = torch.vstack((gene_feature, cell_feature))  # fp32 on the device (DeviceArray slots: no upload)
        g = CellGeneGraph(rowptr, col, val, eid, num_cells + num_feats,
                          {"cell_id": cell_id.to(dev), "feat_id": feat_id.to(dev), "features": features})
        data.data.uns[self.out] = g
        return data


@register_preprocessor("graph", "cell")
class PCACellFeatureGraph(BaseTransform):

    _DISPLAY_ATTRS = ("n_components", "split_name")

    # Where the gene PCA runs.  None: scikit-learn on the host, exactly the reference's call.  "cuda": the exact
    # decomposition on the GPU (dance_amd.utils.pca).  Deliberately NOT a constructor argument — the constructor is the
    # reference's (cell_feature_graph.py:87-97); opt in per instance (``t.pca_device = "cuda"``) or process-wide with
    # DANCE_AMD_PCA_DEVICE.
    pca_device = os.environ.get("DANCE_AMD_PCA_DEVICE") or None

    def __init__(self, n_components: int = 400, split_name=None, *, normalize_edges: bool = True, feat_norm_mode=None,
                 feat_norm_axis: int = 0, mod=None, log_level="WARNING", device="cuda"):
        super().__init__(log_level=log_level)
        self.n_components = n_components
        self.split_name = split_name
        self.normalize_edges = normalize_edges
        self.feat_norm_mode = feat_norm_mode
        self.feat_norm_axis = feat_norm_axis
        self.mod = mod
        self.device = device

    def __call__(self, data):
        WeightedFeaturePCA(self.n_components, self.split_name, feat_norm_mode=self.feat_norm_mode,
                           feat_norm_axis=self.feat_norm_axis, log_level=self.log_level, device=self.pca_device)(data)
        CellFeatureGraph(cell_feature_channel="WeightedFeaturePCA", mod=self.mod, normalize_edges=self.normalize_edges,
                         log_level=self.log_level, device=self.device)(data)
        return data
