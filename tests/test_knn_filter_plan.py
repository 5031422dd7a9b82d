"""CPU: the host-side plan of the kNN filter for d <= 64 (dh_knn_filter_plan, knn_filter.hip::knn_fold_geom) — no device needed.

The candidate rows are grouped by residue class so that each threshold pass reads contiguous tiles; the invariants below are
what the kernels rely on."""
import ctypes

import numpy as np
import pytest

FIELDS = 26


def plan(n, d, nq, k):
    from dance_amd._lib import load
    out = (ctypes.c_int64 * FIELDS)()
    assert load().dh_knn_filter_plan(n, d, nq, k, out, FIELDS) == FIELDS
    f = list(out)
    p = dict(n_pass=f[0], G=f[1], H=f[2], qmagic=f[3], n1=f[4], n_pos=f[5], S=f[6], stride0=f[7], K3=f[8], cap=f[9], tiles=f[10])
    p["passes"] = [dict(t_begin=f[11 + 5 * i], t_end=f[12 + 5 * i], tps=f[13 + 5 * i], n_seg=f[14 + 5 * i], seg=f[15 + 5 * i])
                   for i in range(3)]
    return p


def class_of_slot(slot, G, H, qmagic):
    gh = G // H
    m = slot - H
    return slot * gh if slot < H else m + ((m * qmagic) >> 16) + 1


CASES = [(1000, 50, 1000, 15), (20_000, 8, 20_000, 5), (32_768, 50, 32_768, 15), (40_013, 20, 2_000, 15),
         (262_144, 50, 262_144, 15), (270_011, 12, 1_000, 10), (1_000_000, 50, 1_000_000, 15), (1_000_000, 64, 512, 64),
         (5_000_000, 32, 5_000_000, 30)]


@pytest.mark.parametrize("n,d,nq,k", CASES)
def test_plan_invariants(n, d, nq, k):
    p = plan(n, d, nq, k)
    G, H = p["G"], p["H"]
    assert p["n_pass"] == (3 if n >= 262_144 else 2 if n >= 32_768 else 1)
    assert (G, H) == {1: (1, 1), 2: (16, 1), 3: (64, 8)}[p["n_pass"]]
    # slots <-> residue classes: a bijection; the first H slots are the multiples of G / H (what pass 1 and the middle pass cover)
    classes = [class_of_slot(s, G, H, p["qmagic"]) for s in range(G)]
    assert sorted(classes) == list(range(G))
    assert classes[:H] == [i * (G // H) for i in range(H)]
    # every candidate has exactly one row of the operand: row = slot * n1 + r // G
    n1 = p["n1"]
    assert n1 == -(-n // G) and p["n_pos"] >= G * n1 and p["n_pos"] % (4 * 128) == 0
    slot_of_class = {c: s for s, c in enumerate(classes)}
    r = np.unique(np.concatenate([np.arange(0, min(n, 5000)), np.arange(max(0, n - 5000), n),
                                  np.random.default_rng(0).integers(0, n, 5000)]))
    rows = np.array([slot_of_class[int(c)] for c in r % G]) * n1 + r // G
    assert rows.max() < p["n_pos"] and len(np.unique(rows)) == len(r)
    back = (rows % n1) * G + np.array([classes[int(s)] for s in rows // n1])
    assert np.array_equal(back, r)
    # the sample is a subset of pass 1's rows (class 0) and inside the data
    assert p["stride0"] % G == 0 and (p["S"] - 1) * p["stride0"] < n and p["S"] >= min(n, 1000)
    # passes tile the operand without gaps; pass p ends behind the last row of its classes; ranges are whole LDS images
    assert p["tiles"] * 128 == p["n_pos"]
    t = 0
    ends = {1: [p["tiles"]], 2: [-(-n1 // 128), p["tiles"]], 3: [-(-n1 // 128), -(-(H * n1) // 128), p["tiles"]]}[p["n_pass"]]
    slots = 0
    for i in range(p["n_pass"]):
        ps = p["passes"][i]
        assert ps["t_begin"] == t and ps["t_end"] >= min(ends[i], p["tiles"]) and ps["t_begin"] % 4 == 0
        assert ps["tps"] % 4 == 0 and ps["n_seg"] == -(-(ps["t_end"] - ps["t_begin"]) // ps["tps"]) and ps["seg"] >= 256
        t = ps["t_end"]
        slots += ps["n_seg"] * ps["seg"]
    assert t == p["tiles"] and slots == p["cap"]
    for i in range(p["n_pass"], 3):
        assert p["passes"][i]["n_seg"] == 0
    # few query blocks: the candidates are sliced until >= 512 workgroups exist (or every image has its own slice)
    qblocks = -(-nq // 256)
    if qblocks < 512:
        ps = p["passes"][p["n_pass"] - 1]
        assert qblocks * ps["n_seg"] >= 512 or ps["tps"] == 4
    # operand width: d padded to 8, six threshold columns, whole 16-wide MFMA steps
    assert p["K3"] == ((d + 7) // 8 * 8 + 6 + 15) // 16 * 16


def test_tile_kernel_has_no_plan():
    assert plan(100_000, 2000, 100_000, 15)["n_pass"] == 0


def test_workspace_covers_the_plan():
    from dance_amd._lib import load
    n, d, nq, k = 1_000_000, 50, 1_000_000, 15
    p = plan(n, d, nq, k)
    ws = load().dh_knn_bruteforce_f32_workspace_bytes(n, d, nq, k, 2)
    need = nq * p["cap"] * 4 + (n + p["n_pos"]) * p["K3"] * 2 + n * 52 * 4
    assert need <= ws <= need + (64 << 20)
