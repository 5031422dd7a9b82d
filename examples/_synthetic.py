"""Synthetic stand-ins for the datasets the reference's examples download (there is no network here): count matrices drawn from
cell-type-specific expression programmes, spot coordinates with spatially coherent domains, a histology image.  Every example builds
a ``dance_amd.data.Data`` the way ``dataloader.load_data(transform=...)`` would and hands the count matrix over as a device slot."""
import numpy as np
import pandas as pd
import torch


def counts(n_cells: int, n_genes: int, n_types: int, seed: int):
    rng = np.random.default_rng(seed)
    types = rng.integers(0, n_types, n_cells)
    rates = rng.gamma(0.3, 1.0, (n_types, n_genes)) * 2 + 0.02
    x = rng.poisson(rates[types] * rng.uniform(0.6, 1.6, (n_cells, 1))).astype(np.float32)
    x[np.arange(n_cells), rng.integers(0, n_genes, n_cells)] += 1  # no empty cell
    return x, types


def spots(n_side: int, n_domains: int, n_genes: int, seed: int):
    """A n_side x n_side grid of spots whose domain is a vertical stripe; counts follow the domain's programme."""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    xy = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64)
    domain = np.minimum((xy[:, 0] * n_domains / n_side).astype(int), n_domains - 1)
    rates = rng.gamma(0.4, 1.0, (n_domains, n_genes)) * 3 + 0.05
    x = rng.poisson(rates[domain]).astype(np.float32)
    x[np.arange(len(x)), rng.integers(0, n_genes, len(x))] += 1
    xy_pixel = (xy * 6 + 10).astype(np.int64)
    image = rng.integers(0, 255, (n_side * 6 + 30, n_side * 6 + 30, 3)).astype(np.uint8)
    return x, domain, xy, xy_pixel, image


def as_data(x, *, device="cuda", obs=None, obsm=None, uns=None, var_names=None, train_size="all", **split):
    from dance_amd.data import AnnDataLite, Data, DeviceArray
    slot = DeviceArray(torch.from_numpy(x).to(device)) if device is not None else x
    var = None if var_names is None else pd.DataFrame(index=list(var_names))
    obs = None if obs is None else pd.DataFrame(obs, index=[str(i) for i in range(x.shape[0])])
    return Data(AnnDataLite(slot, obs=obs, var=var, obsm=obsm, uns=uns), train_size=train_size, **split)
