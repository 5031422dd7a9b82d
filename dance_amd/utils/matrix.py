"""dance.utils.matrix on MI355X: ``pairwise_distance`` (numba kernel of dance/utils/matrix.py:164-180) runs as the HIP
kernel dh_pairwise_distance_f32; ``normalize`` (:8-67) keeps its numpy/torch semantics (it is host-side feature
prep, not a kernel)."""
import numpy as np
import torch

from .. import kernels

DIST_FUNC_ID = ["euclidean_distance", "pearson_distance", "spearman_distance"]


def pairwise_distance(x, dist_func_id: int = 0, *, device="cuda", return_device: bool = False):
    """All-pairs distance matrix [n, n] f32.  ``x``: f32 ndarray (as the reference requires, f4[:,:]) or GPU tensor.

    Returns an ndarray like the reference unless ``return_device`` — the N^2 result then stays in HBM."""
    if dist_func_id not in (0, 1, 2):
        raise ValueError("Unknown distance function ID")
    if isinstance(x, np.ndarray):
        if x.dtype != np.float32:
            raise TypeError(f"pairwise_distance expects float32 input (numba signature f4[:,:]), got {x.dtype}")
        xt = torch.from_numpy(np.ascontiguousarray(x)).to(device)
    else:
        xt = x
    out = kernels.pairwise_distance(xt, dist_func_id)
    return out if return_device else out.cpu().numpy()


def normalize(mat, *, mode="normalize", axis: int = 0, eps: float = -1.0):
    """Same contract as dance/utils/matrix.py:8-67 for ndarray or torch tensor."""
    if isinstance(mat, torch.Tensor):
        is_torch = True
    elif not isinstance(mat, np.ndarray):
        raise TypeError(f"Invalid type for input matrix: {type(mat)}")
    else:
        is_torch = False
    opts = {"axis": axis, "keepdims": True}
    shift, denom = 0, None
    if mode == "standardize":
        shift = -mat.mean(**opts)
        denom = mat.std(**opts, unbiased=False) if is_torch else mat.std(**opts)
    elif mode == "minmax":
        lo = mat.min(**opts)[0] if is_torch else mat.min(**opts)
        hi = mat.max(**opts)[0] if is_torch else mat.max(**opts)
        shift, denom = -lo, hi - lo
    elif mode == "normalize":
        denom = mat.sum(**opts)
    elif mode == "l2":
        denom = (mat**2).sum(**opts)**0.5
    if denom is None:
        denom = 1
    elif eps == -1:
        denom[denom == 0] = 1
    elif eps > 0:
        denom = denom + eps
    else:
        raise ValueError(f"Invalid {eps=!r}. Must be positive or -1, the later set zero entries to one.")
    return (mat + shift) / denom
