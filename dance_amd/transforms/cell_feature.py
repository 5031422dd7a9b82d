"""Cell / gene PCA features feeding the graph builders (dance/transforms/cell_feature.py:19-75,146-194).
``device=None``: scikit-learn on the host, as in the reference.  ``device="cuda"``: the decomposition on the GPU
(dance_amd/utils/pca.py) with inputs and outputs kept on the device as ``DeviceArray``s."""
import numpy as np
from sklearn.decomposition import PCA

from ..registry import register_preprocessor
from ..utils.matrix import normalize
from .base import BaseTransform


@register_preprocessor("feature", "cell")
class WeightedFeaturePCA(BaseTransform):
    """Gene PCA on the (train-split) expression matrix; cell feature = row-normalised X @ gene features."""

    _DISPLAY_ATTRS = ("n_components", "split_name", "feat_norm_mode", "feat_norm_axis")

    # solver of the device path (attributes, so the constructor stays the reference's): "full" = exact decomposition
    # (dance_amd.utils.pca.pca_scores); "randomized" = the solver sklearn's "auto" picks for these shapes, seeded by
    # ``device_random_state`` through the same numpy stream (pca_scores_randomized) — reproduces a seeded host run
    device_solver = "full"
    device_random_state = None

    def __init__(self, n_components=400, split_name=None, feat_norm_mode=None, feat_norm_axis=0, save_info=False, *,
                 device=None, **kwargs):
        super().__init__(**kwargs)
        self.n_components = n_components
        self.split_name = split_name
        self.feat_norm_mode = feat_norm_mode
        self.feat_norm_axis = feat_norm_axis
        self.save_info = save_info
        # device=None: scikit-learn on the host, exactly the reference's call.  device="cuda": the exact decomposition on
        # the GPU (dance_amd.utils.pca) — an explicit opt-in, because sklearn's default solver for these shapes is a
        # randomised SVD that the exact result only matches to that solver's accuracy.
        self.device = device

    def __call__(self, data):
        if self.device is not None:
            return self._call_on_device(data)
        feat = data.get_x(self.split_name)  # cells x genes
        if self.feat_norm_mode is not None:
            feat = normalize(feat, mode=self.feat_norm_mode, axis=self.feat_norm_axis)
        if self.n_components > min(feat.shape):
            self.logger.warning(f"n_components={self.n_components} must be between 0 and "
                                f"min(n_samples, n_features)={min(feat.shape)} with svd_solver='full'")
            self.n_components = min(feat.shape)
        gene_pca = PCA(n_components=self.n_components)
        gene_feat = gene_pca.fit_transform(feat.T)  # genes x components
        x = data.get_x()
        cell_feat = normalize(x, mode="normalize", axis=1) @ gene_feat
        data.data.obsm[self.out] = cell_feat.astype(np.float32)
        data.data.varm[self.out] = gene_feat.astype(np.float32)
        if self.save_info:
            data.data.uns["pca_components"] = gene_pca.components_
            data.data.uns["pca_mean"] = gene_pca.mean_
            data.data.uns["pca_explained_variance"] = gene_pca.explained_variance_
            data.data.uns["pca_explained_variance_ratio"] = gene_pca.explained_variance_ratio_
        return data


    def _call_on_device(self, data):
        """Same transform with every matrix on the device: X is read as a device tensor (no copy if an earlier device
        transform left a DeviceArray there), the results are DeviceArrays in obsm / varm — nothing crosses PCIe."""
        import torch

        from .. import kernels
        from ..data import DeviceArray
        from ..utils.pca import pca_scores, pca_scores_randomized
        if self.device_solver not in ("full", "randomized"):
            raise ValueError(f"device_solver must be 'full' or 'randomized', got {self.device_solver!r}")
        if self.save_info and self.device_solver == "full":
            raise NotImplementedError("save_info needs the cell-space components, which the exact device path never forms")
        feat = data.get_x(self.split_name, return_type=self.device)  # cells x genes, on the device
        if self.feat_norm_mode is not None:
            feat = normalize(feat, mode=self.feat_norm_mode, axis=self.feat_norm_axis)
        if self.n_components > min(feat.shape):
            self.logger.warning(f"n_components={self.n_components} must be between 0 and "
                                f"min(n_samples, n_features)={min(feat.shape)} with svd_solver='full'")
            self.n_components = min(feat.shape)
        ft = feat.t().contiguous()  # genes x cells
        if self.device_solver == "randomized":
            gene_feat, comps, var = pca_scores_randomized(ft, self.n_components, self.device_random_state)
            if self.save_info:
                data.data.uns["pca_components"] = comps.cpu().numpy()
                data.data.uns["pca_mean"] = ft.mean(0).cpu().numpy()
                data.data.uns["pca_explained_variance"] = var.cpu().numpy()
                total = ft.var(0, unbiased=True).sum() if ft.shape[0] > 1 else ft.new_zeros(())
                data.data.uns["pca_explained_variance_ratio"] = (var / total).cpu().numpy()
        else:
            gene_feat, _, _ = pca_scores(ft, self.n_components)                                  # genes x components
        x = data.get_x(return_type=self.device)
        cell_feat = kernels.gemm(normalize(x, mode="normalize", axis=1).contiguous(), gene_feat.contiguous())
        data.data.obsm[self.out] = DeviceArray(cell_feat)
        data.data.varm[self.out] = DeviceArray(gene_feat.contiguous())
        return data


@register_preprocessor("feature", "cell")
class CellPCA(BaseTransform):
    """PCA of the cell feature matrix -> obsm[out]."""

    _DISPLAY_ATTRS = ("n_components", )

    def __init__(self, n_components=400, *, channel=None, mod=None, save_info=False, svd_solver="auto", device=None, **kwargs):
        super().__init__(**kwargs)
        self.n_components = n_components
        self.channel = channel
        self.save_info = save_info
        self.svd_solver = svd_solver
        # None: scikit-learn on the host (reference); "cuda": PCA on the GPU (opt-in) — exact for svd_solver "auto" / "full" /
        # "covariance_eigh", sklearn's randomised solver restated for svd_solver="randomized" (seed: ``device_random_state``)
        self.device = device
        self.device_random_state = None

    def __call__(self, data):
        if self.device is not None:
            from ..data import DeviceArray
            from ..utils.pca import pca_scores, pca_scores_randomized
            xd = data.get_feature(return_type=self.device, channel=self.channel, channel_type="obsm" if self.channel else "X").contiguous()
            if self.n_components > min(xd.shape):
                self.n_components = min(xd.shape)
            if self.svd_solver == "randomized":
                scores, comps, _ = pca_scores_randomized(xd, self.n_components, self.device_random_state)
            elif self.svd_solver in ("auto", "full", "covariance_eigh"):
                scores, comps, _ = pca_scores(xd, self.n_components)
            else:
                raise ValueError(f"svd_solver={self.svd_solver!r} has no device path (use 'auto', 'full' or 'randomized')")
            data.data.obsm[self.out] = DeviceArray(scores.contiguous())
            if self.save_info:
                if comps is None:
                    raise NotImplementedError("save_info with more features than samples: components are not formed on the device")
                data.data.uns["pca_components"] = comps.cpu().numpy()
                data.data.uns["pca_mean"] = xd.mean(0).cpu().numpy()
            return data
        feat = data.get_feature(return_type="numpy", channel=self.channel, channel_type="obsm" if self.channel else "X")
        if self.n_components > min(feat.shape):
            self.n_components = min(feat.shape)
        pca = PCA(n_components=self.n_components, svd_solver=self.svd_solver)
        data.data.obsm[self.out] = pca.fit_transform(feat)
        if self.save_info:
            data.data.uns["pca_components"] = pca.components_
            data.data.uns["pca_mean"] = pca.mean_
        return data
