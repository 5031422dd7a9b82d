"""CPU: the error budget of the kNN filter (dance_amd/csrc/knn_filter.hip, head of the file), checked numerically for both forms:
d <= 64 (one fp16 term, thresholds folded into the matrix-core operands, sign-bit test) and d > 64 (three bf16 terms, one fma +
compare per pair).

The filter may never drop a true neighbour: for every pair (q, c) whose chain distance is <= the query's threshold tau, the
matrix-core accumulator  -2 yh_q . yh_c + Cn[c] - Rq[q]  must come out NEGATIVE.  The hardest case is the boundary
tau == d2_chain(q, c), which is what is tested here, pair by pair, with the device's formulas restated in numpy (fp16 roundings
by numpy's float16 = round-to-nearest-even; both with subnormals and with everything below fp16's normal range flushed to zero;
the fp32 accumulation of the matrix cores replaced by exact float64 sums plus its worst-case rounding allowance).
No oracle/ import needed: the chain is three lines."""
import numpy as np
import pytest

F32 = np.float32
MUL = F32(4096.0)


def _chain_d2(xq, xc):
    """d2 = (((0 + rn((x0-y0)^2)) + ...): every op a separate float32 rounding, features in order (knn.hip's definition)."""
    acc = np.zeros(xq.shape[0], dtype=F32)
    for t in range(xq.shape[1]):
        diff = (xq[:, t] - xc[:, t]).astype(F32)
        acc = (acc + (diff * diff).astype(F32)).astype(F32)
    return acc


def _f16(v, ftz):
    h = v.astype(np.float16)
    if ftz:
        h = np.where(np.abs(v) < 2.0 ** -14, np.float16(0), h)
    return h


def _split3h(w, ftz):
    t0 = _f16(w, ftz)
    r1 = (w - t0.astype(F32)).astype(F32)
    t1 = _f16(r1, ftz)
    t2 = _f16((r1 - t1.astype(F32)).astype(F32), ftz)
    return t0.astype(np.float64) + t1.astype(np.float64) + t2.astype(np.float64)


def _bf16(v):
    u = v.astype(F32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(F32)


def _tile_form(x, qi, ci):
    """d > 64: centred rows split in three bf16 terms, fma(-2, dot, Cn) <= Rq.  Returns (lhs - rhs, allowance)."""
    n, d = x.shape
    u = 2.0 ** -24
    eps = F32(2.0 ** -14 + d * 2.0 ** -22)
    chain_rel = F32((d + 16) * 6.3e-8)
    mu = (x.sum(axis=0, dtype=np.float64) / n).astype(F32)
    xc = (x - mu).astype(F32)
    hi = _bf16(xc)
    lo = _bf16((xc - hi).astype(F32))
    nrm = (xc.astype(np.float64) ** 2).sum(axis=1).astype(F32)   # accumulated in double on the device too
    cn = ((F32(1) - eps) * nrm).astype(F32)
    tau = _chain_d2(x[qi], x[ci])
    rq = ((tau * chain_rel).astype(np.float64) + tau).astype(F32)  # fma: one rounding
    rq = (rq - ((F32(1) - eps) * nrm[qi]).astype(F32)).astype(F32)
    hq, lq, hc, lc = (a.astype(np.float64) for a in (hi[qi], lo[qi], hi[ci], lo[ci]))
    dot = (hq * hc + hq * lc + lq * hc).sum(axis=1)
    mag = (np.abs(hq * hc) + np.abs(hq * lc) + np.abs(lq * hc)).sum(axis=1)
    acc = -2.0 * dot + cn[ci].astype(np.float64) - rq.astype(np.float64)
    # fp32 accumulation over 3 d exact products: 3 d u (1 + 2^-7) sum |terms|, 1.3x that allowed for; the fma adds one rounding
    allowance = 2.0 * 1.3 * 3 * d * u * (1 + 2.0 ** -7) * mag + 4 * u * (2 * np.abs(dot) + np.abs(cn[ci]))
    return acc, allowance


def _accumulators(x, qi, ci, ftz):
    """Accumulator of pair (qi[j], ci[j]) with tau = the pair's own chain distance, and the allowance for fp32 accumulation."""
    n, d = x.shape
    if d > 64:
        return _tile_form(x, qi, ci)
    fold = True
    dp = (d + 7) // 8 * 8
    eps = F32(2.0 ** -10 + 2.0 ** -13 + dp * 2.0 ** -20)
    abs_lin = F32(1.01 * 2.0 ** -13 * np.sqrt(dp))
    mu = (x.sum(axis=0, dtype=np.float64) / n).astype(F32)
    xc = (x - mu).astype(F32)
    m = float(np.abs(xc).max())
    assert 2.0 ** -60 <= m <= 2.0 ** 60
    s = F32(2.0 ** (8 - int(np.floor(np.log2(m)))))
    y = (xc * s).astype(F32)                                   # exact
    assert 256.0 <= float(np.abs(y).max()) < 512.0
    yh = _f16(y, ftz)
    assert np.isfinite(yh.astype(F32)).all()
    nrm = (y.astype(np.float64) ** 2).sum(axis=1).astype(F32)  # the device sums in fp32: (d + 2) u, inside the slack
    cn = (((F32(1) - eps) * nrm).astype(F32) - (abs_lin * np.sqrt(nrm).astype(F32)).astype(F32)).astype(F32)
    cn_sum = _split3h((cn * F32(1.0 / 4096.0)).astype(F32), ftz) * 4096.0 if fold else cn.astype(np.float64)
    tau = _chain_d2(x[qi], x[ci])                              # boundary: the candidate IS the k-th neighbour
    tau_s = ((tau * s).astype(F32) * s).astype(F32)
    nq2 = nrm[qi]
    rq = (tau_s * F32(1.0 + 2.0 ** -20)).astype(F32)
    rq = (rq - ((F32(1) - eps) * nq2).astype(F32)).astype(F32)
    rq = (rq + (abs_lin * np.sqrt(nq2).astype(F32)).astype(F32)).astype(F32)
    rq = (rq + F32(1)).astype(F32)
    if fold:
        rq = np.minimum(rq, F32(2.0 ** 26))
        rq_sum = _split3h((-rq * F32(1.0 / 4096.0)).astype(F32), ftz) * 4096.0
    else:
        rq_sum = -rq.astype(np.float64)
    a = (-2.0 * yh[qi].astype(np.float64))
    b = yh[ci].astype(np.float64)
    dot = (a * b).sum(axis=1)
    acc = dot + cn_sum[ci] + rq_sum
    # fp32 accumulation of K = dp + 6 exact products in any order: <= K u (1 + 2^-7) sum |terms| (twice that allowed for)
    if fold:
        terms = np.abs(a * b).sum(axis=1) + np.abs(cn_sum[ci]) + np.abs(rq_sum)
        allowance = 2.0 * (dp + 6) * 2.0 ** -24 * (1 + 2.0 ** -7) * terms
    else:  # the dot product accumulates on the matrix cores; fma(-2, dot, Cn) <= Rq adds two roundings
        allowance = 2.0 * d * 2.0 ** -24 * (1 + 2.0 ** -7) * np.abs(a * b).sum(axis=1) + 4 * 2.0 ** -24 * (np.abs(dot) + np.abs(cn_sum[ci]))
    return acc, allowance


def _pairs(rng, x, n_pairs):
    n = x.shape[0]
    qi = rng.integers(0, n, size=n_pairs)
    ci = rng.integers(0, n, size=n_pairs)
    ci[: n_pairs // 8] = qi[: n_pairs // 8]                    # the query itself (d2 = 0)
    # near pairs: for a slice of the queries the nearest other point by float64 distance
    sel = slice(n_pairs // 8, n_pairs // 4)
    sub = x[qi[sel]].astype(np.float64)
    for j, row in enumerate(sub):
        dd = ((x.astype(np.float64) - row) ** 2).sum(axis=1)
        dd[qi[sel][j]] = np.inf
        ci[n_pairs // 8 + j] = int(np.argmin(dd))
    return qi, ci


def _dataset(kind, rng):
    n = 3000
    if kind == "normal50":
        return rng.standard_normal((n, 50))
    if kind == "spread_clusters":      # within-cluster sigma 1, centres 30 apart per axis
        return rng.standard_normal((n, 50)) + rng.standard_normal((25, 50))[rng.integers(0, 25, n)] * 30.0
    if kind == "far_offset":
        return rng.standard_normal((n, 24)) + 1000.0
    if kind == "integers":
        return rng.integers(-3, 4, size=(n, 6)).astype(np.float64)
    if kind == "tiny":
        return rng.standard_normal((n, 33)) * 1e-12
    if kind == "huge":
        return rng.standard_normal((n, 33)) * 1e12
    if kind == "heavy_tail":           # 18+ binades between the largest and the typical element: small ones fall below fp16's range
        return rng.standard_normal((n, 40)) * np.exp(rng.standard_normal((n, 40)) * 4.0)
    if kind == "one_outlier":          # a single huge row sets the scale; everything else is tiny next to it
        x = rng.standard_normal((n, 16)) * 1e-3
        x[7] = 1e4
        return x
    if kind == "duplicates":
        x = rng.standard_normal((n, 10))
        x[::3] = x[0]
        return x
    if kind == "d64":
        return rng.standard_normal((n, 64)) * rng.uniform(0.01, 100.0, size=(1, 64))
    if kind == "d3":
        return rng.random((n, 3)) * 100.0
    if kind == "d130":                 # the tile kernel's form (d > 64)
        return rng.standard_normal((n, 130)) + rng.standard_normal((12, 130))[rng.integers(0, 12, n)] * 10.0
    if kind == "d2000":                # raw expression-like rows: non-negative, sparse
        return np.maximum(rng.standard_normal((600, 2000)) * 2.0 - 2.0, 0.0)
    raise ValueError(kind)


@pytest.mark.parametrize("ftz", [False, True])
@pytest.mark.parametrize("kind", ["normal50", "spread_clusters", "far_offset", "integers", "tiny", "huge", "heavy_tail",
                                  "one_outlier", "duplicates", "d64", "d3", "d130", "d2000"])
def test_boundary_pairs_always_pass(kind, ftz):
    rng = np.random.default_rng(sum(map(ord, kind)))
    x = _dataset(kind, rng).astype(F32)
    qi, ci = _pairs(rng, x, 4000 if x.shape[1] <= 130 else 1200)
    acc, allowance = _accumulators(x, qi, ci, ftz)
    worst = (acc + allowance).max()
    assert worst <= 0.0, (kind, ftz, float(worst))


def test_the_slack_is_not_vacuous():
    """The same accumulator with a threshold well INSIDE the pair's distance is positive: the filter does discard pairs."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3000, 50)).astype(F32)
    qi = rng.integers(0, 3000, size=2000)
    ci = (qi + 1 + rng.integers(0, 2998, size=2000)) % 3000
    acc, allowance = _accumulators(x, qi, ci, False)
    # tau = d2 gives acc in (-margin, 0); the margin is ~ 2 eps (|q|^2 + |c|^2) ~ 2.3e-3 of the norms: tiny against d2 itself
    n, d = x.shape
    y_scale = 2.0 ** (8 - int(np.floor(np.log2(float(np.abs(x - x.mean(0)).max())))))
    d2_scaled = _chain_d2(x[qi], x[ci]).astype(np.float64) * y_scale ** 2
    assert np.all(acc < 0) and np.median(-acc / d2_scaled) < 0.01
