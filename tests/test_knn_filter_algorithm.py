"""CPU: the staged-threshold algorithm of the kNN filter (knn_filter.hip, d <= 64) walked through in numpy on the plan the library
itself reports (dh_knn_filter_plan): rows grouped by residue class, thresholds from a strided sample, then pass by pass
"filter with a superset test -> exact k smallest of the survivors merged into the keys so far -> next threshold".

What it pins down, without a device: with ANY superset filter (here: the exact distance with a random positive slack) the result
is the exact k nearest by (d2, index); queries whose survivor segment overflows re-scan everything and stay exact; keys seen twice
(a re-scan in an early pass carries candidates of later passes) collapse.  The kernels are checked against the scan on the GPU
(tests/test_gpu_graphs.py); this is the host logic they implement."""
import ctypes

import numpy as np
import pytest


def _plan(n, d, nq, k):
    from dance_amd._lib import load
    out = (ctypes.c_int64 * 26)()
    load().dh_knn_filter_plan(n, d, nq, k, out, 26)
    f = list(out)
    return dict(n_pass=f[0], G=f[1], H=f[2], qmagic=f[3], n1=f[4], n_pos=f[5], S=f[6], stride0=f[7],
                passes=[dict(t_begin=f[11 + 5 * i], t_end=f[12 + 5 * i], n_seg=f[14 + 5 * i], seg=f[15 + 5 * i]) for i in range(3)])


def _candidate_of_row(p, plan, n):
    """candidate id of operand row p (-1: a zero row)"""
    G, H, n1 = plan["G"], plan["H"], plan["n1"]
    slot, j = p // n1, p % n1
    m = slot - H
    cls = np.where(slot < H, slot * (G // H), m + ((m * plan["qmagic"]) >> 16) + 1)
    r = j * G + cls
    return np.where((p < G * n1) & (r < n), r, -1)


def _keys(d2, idx):
    """(bits(d2) << 32 | index): unsigned order = (d2, index) order for d2 >= +0"""
    return (d2.astype(np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | idx.astype(np.uint64)


def _k_smallest(keys, k):
    keys = np.unique(keys)  # equal keys collapse (the device's "smallest key above the previous pick")
    return keys[:k]


def staged_knn(x, k, plan, rng, force_overflow=()):
    n = x.shape[0]
    d2_all = ((x[:, None, :].astype(np.float64) - x[None, :, :].astype(np.float64)) ** 2).sum(-1).astype(np.float32)
    sample = np.arange(plan["S"]) * plan["stride0"]
    assert sample.max() < n and np.all(sample % plan["G"] == 0)
    out = np.zeros((n, k), dtype=np.uint64)
    rows = np.arange(plan["n_pos"])
    cand = _candidate_of_row(rows, plan, n)
    for q in range(n):
        d2 = d2_all[q]
        tau = np.sort(d2[sample])[k - 1] if len(sample) >= k else np.inf
        carried = np.zeros(0, dtype=np.uint64)
        for ps in range(plan["n_pass"]):
            P = plan["passes"][ps]
            c = cand[P["t_begin"] * 128:P["t_end"] * 128]
            c = c[c >= 0]
            passes = d2[c] <= tau * (1.0 + rng.uniform(0.0, 0.02)) + 1e-30     # any superset of {d2 <= tau}
            surv = c[passes]
            if len(surv) > P["n_seg"] * P["seg"] or (q, ps) in force_overflow:   # a segment over capacity: re-scan everything
                surv = np.arange(n)
            carried = _k_smallest(np.concatenate([carried, _keys(d2[surv], surv)]), k)
            tau = np.uint32(carried[k - 1] >> np.uint64(32)).view(np.float32) if len(carried) >= k else np.inf
        out[q] = carried
    ref = np.stack([_k_smallest(_keys(d2_all[q], np.arange(n)), k) for q in range(n)])
    return out, ref


@pytest.mark.parametrize("shape", ["one_pass", "two_pass", "three_pass"])
def test_staged_thresholds_are_exact(shape):
    rng = np.random.default_rng(11)
    k, d = 7, 5
    n_real = {"one_pass": 20_000, "two_pass": 40_013, "three_pass": 270_011}[shape]
    plan = _plan(n_real, d, n_real, k)
    assert plan["n_pass"] == {"one_pass": 1, "two_pass": 2, "three_pass": 3}[shape]
    # the walk-through is O(n^2) in numpy: shrink the plan to a toy size with the same class structure (G, H, slot order), 128-row
    # tiles and the pass boundaries the geometry code would give it
    G, H = plan["G"], plan["H"]
    n = 1500 if G > 1 else 600
    n1 = -(-n // G)
    n_pos = -(-(G * n1) // 512) * 512
    tiles = n_pos // 128
    t1 = min(tiles, -(-(-(-n1 // 128)) // 4) * 4)
    t2 = min(tiles, -(-(-(-(H * n1) // 128)) // 4) * 4)
    bounds = {1: [0, tiles], 2: [0, t1, tiles], 3: [0, t1, t2, tiles]}[plan["n_pass"]]
    stride0 = G * 2 if G > 1 else 3
    toy = dict(plan, n1=n1, n_pos=n_pos, stride0=stride0, S=-(-n // stride0),
               passes=[dict(t_begin=bounds[i], t_end=bounds[i + 1], n_seg=1, seg=64 if i else 200) for i in range(plan["n_pass"])])
    # every candidate owns exactly one operand row
    c = _candidate_of_row(np.arange(n_pos), toy, n)
    assert np.array_equal(np.sort(c[c >= 0]), np.arange(n))
    x = rng.integers(-4, 5, size=(n, d)).astype(np.float32)       # many exact ties: the index order matters
    x[::7] = x[0]                                                 # and duplicates
    got, ref = staged_knn(x, k, toy, rng, force_overflow={(3, 0), (5, toy["n_pass"] - 1), (9, 0), (9, toy["n_pass"] - 1)})
    assert np.array_equal(got, ref)
