"""CPU restatement of ONE training step of the reference's two mini-batch loops, with the dropout masks of csrc/ministep.hip made explicit.

TEST INFRASTRUCTURE ONLY (imported by tests/; nothing under dance_amd/ imports it).  float64 torch / numpy, the REFERENCE's operation
order (graph-sc multiplies by W first and aggregates at width ``hidden``: graphsc.py:452-465 — the kernels aggregate first; the two agree
to rounding and this file is what says so).  Every block cites the reference lines it follows (paths relative to /root/reference):

* ``philox4x32_10`` — the Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
  SC'11; the generator behind torch's CUDA dropout).  Pinned by the Random123 known-answer vectors (tests/test_oracle_ministep.py).
* ``dropout_mask`` — the keying of ministep.hip's ``Drop``: element e of stream sid at step t is kept iff the top 24 bits of word (e & 3)
  of philox(counter = (e >> 2, sid, t_lo, t_hi), key = (seed_lo, seed_hi)) are below floor((1 - p) 2^24); kept values scale by 1 / (1 - p).
* ``graphsc_step`` — dance/modules/single_modality/clustering/graphsc.py:196-219 (one batch of ``GraphSC.fit``), :352-383 (GCNAE.forward),
  :405-411 (InnerProductDecoder), :428-484 (WeightedGraphConv).
* ``scdeepsort_step`` — dance/modules/single_modality/cell_type_annotation/scdeepsort.py:238-246, :66-88 (GNN.forward),
  dance/models/nn/gnn.py:62-96 (AdaptiveSAGE: the output ignores ``neigh``).
* ``adam_update`` — torch.optim.Adam (graphsc.py:189, scdeepsort.py:160), single-tensor form.
"""
import numpy as np
import torch
import torch.nn.functional as F

SID_GENE, SID_SELF, SID_DEC, SID_SDS = 0, 2, 4, 5
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """ctr: uint32 [..., 4]; key: two uint32.  Returns uint32 [..., 4]."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for r in range(10):
        if r:
            k0 = (k0 + np.uint64(_W0)) & mask
            k1 = (k1 + np.uint64(_W1)) & mask
        p0, p1 = np.uint64(_M0) * c[0], np.uint64(_M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
    return np.stack(c, -1).astype(np.uint32)


def dropout_mask(n: int, p: float, seed: int, step: int, sid: int) -> np.ndarray:
    """float64 [n]: 0 or 1 / (1 - p) per element (all ones for p == 0)."""
    if p <= 0:
        return np.ones(n)
    quads = (n + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    ctr = np.stack([(q & np.uint64(0xFFFFFFFF)).astype(np.uint32), (np.uint32(sid) | ((q >> np.uint64(32)).astype(np.uint32) << np.uint32(8))),
                    np.full(quads, step & 0xFFFFFFFF, np.uint32), np.full(quads, (step >> 32) & 0xFFFFFFFF, np.uint32)], -1)
    words = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)[:n]
    keep = 1.0 - float(np.float32(p))
    thr = int(keep * 16777216.0)
    return np.where((words >> np.uint32(8)) < thr, float(np.float32(1.0 / keep)), 0.0)


def adam_update(p, g, m, v, t, lr, beta1, beta2, eps, weight_decay=0.0):
    """One Adam step (amsgrad off) on float64 copies; returns (p, m, v).  ``t`` = the step count AFTER this step."""
    if weight_decay:
        g = g + weight_decay * p
    m = m + (1 - beta1) * (g - m)
    v = beta2 * v + (1 - beta2) * g * g
    denom = v.sqrt() / np.sqrt(1 - beta2**t) + eps
    return p - (lr / (1 - beta1**t)) * (m / denom), m, v


def _block(rowptr, col, val, seeds, n_genes):
    """The full-neighbour block of seed CELLS over sources [seeds | all genes] (a superset of dgl.to_block's: genes no seed expresses
    have no edge): (e_src, e_dst, w) with the self loop's source = the seed's own position."""
    es, ed, w = [], [], []
    for i, v in enumerate(seeds):
        for e in range(rowptr[v], rowptr[v + 1]):
            c = col[e]
            es.append(len(seeds) + c if c < n_genes else i)
            assert c < n_genes or c == v
            ed.append(i)
            w.append(val[e])
    return torch.tensor(es), torch.tensor(ed), torch.tensor(np.asarray(w, dtype=np.float64))


def graphsc_step(params, rowptr, col, val, features, n_genes, seeds, *, dropout=0.0, decoder_dropout=0.0, seed=0, step=0, agg="sum"):
    """params: dict W1 [F, H], b1 [H], W2 [E, H], b2 [E] (float64).  Returns (loss, emb of the first forward, dict of gradients)."""
    b, g = len(seeds), n_genes
    feats = torch.as_tensor(features, dtype=torch.float64)
    f = feats.shape[1]
    es, ed, w = _block(rowptr, col, val, seeds, g)
    x_src = torch.cat((feats[torch.as_tensor(np.asarray(seeds))], feats[:g]))
    out_deg = torch.bincount(es, minlength=b + g).double().clamp(min=1)   # degrees INSIDE the block (graphsc.py:444-446)
    in_deg = torch.bincount(ed, minlength=b).double().clamp(min=1)       # :468
    p = {k: torch.as_tensor(v, dtype=torch.float64).clone().requires_grad_(True) for k, v in params.items()}

    def forward(k):
        m_self = torch.from_numpy(dropout_mask(b * f, dropout, seed, step, SID_SELF + k)).reshape(b, f)
        m_gene = torch.from_numpy(dropout_mask(g * f, dropout, seed, step, SID_GENE + k)).reshape(g, f)
        x = x_src * torch.cat((m_self, m_gene))                                       # GCNAE.dropout on the block input (:366-367)
        feat = (x * out_deg.pow(-0.5)[:, None]) @ p["W1"]                             # :447-458 (W first)
        h = torch.zeros((b, feat.shape[1]), dtype=torch.float64).index_add_(0, ed, feat[es] * w[:, None])  # :430-438, :462-463
        if agg == "mean":
            h = h / torch.bincount(ed, minlength=b).clamp(min=1)[:, None]            # :464-465
        h = F.relu(h * in_deg.pow(-0.5)[:, None] + p["b1"])                          # :467-483
        return h @ p["W2"].t() + p["b2"]                                              # the hidden Linear (:376-377)

    with torch.no_grad():
        emb1 = forward(0)                                                              # graphsc.py:202-203
    emb2 = forward(1)                                                                  # :215
    zd = emb2 * torch.from_numpy(dropout_mask(b * emb2.shape[1], decoder_dropout, seed, step, SID_DEC)).reshape(emb2.shape)  # :409
    logits = zd @ zd.t()                                                               # :410 (identity activation)
    adj = torch.eye(b, dtype=torch.float64)                                            # :208-209: the seeds' self loops
    tot = float(adj.sum())
    pos_weight = torch.tensor([(b * b - tot) / tot], dtype=torch.float64)             # :210
    factor = (b * b - tot) * 2 or 1                                                    # :211-213
    loss = (b * b / factor) * F.binary_cross_entropy_with_logits(logits, adj, pos_weight=pos_weight)  # :214-216
    loss.backward()
    return float(loss.detach()), emb1.numpy(), {k: v.grad.numpy() for k, v in p.items()}


def scdeepsort_step(params, features, labels, seeds, *, dropout=0.0, seed=0, step=0):
    """params: W1 [H, D], b1 [H], W2 [C, H], b2 [C].  The layer output is Linear(dropout(h_dst)) (gnn.py:92-96); summed cross entropy
    (scdeepsort.py:185,243).  Returns (loss, dict of gradients)."""
    sd = torch.as_tensor(np.asarray(seeds))
    x = torch.as_tensor(features, dtype=torch.float64)[sd]
    b, d = x.shape
    p = {k: torch.as_tensor(v, dtype=torch.float64).clone().requires_grad_(True) for k, v in params.items()}
    z = x * torch.from_numpy(dropout_mask(b * d, dropout, seed, step, SID_SDS)).reshape(b, d)
    h1 = F.relu(z @ p["W1"].t() + p["b1"])
    logits = h1 @ p["W2"].t() + p["b2"]
    loss = F.cross_entropy(logits, torch.as_tensor(labels)[sd], reduction="sum")
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in p.items()}


def sage_neigh(rowptr, col, val, features, cell_id, alpha, n_genes, seeds):
    """gnn.py:62-90: mean over the in-edges of alpha[idx(e)] * w_e * h[src(e)] for seed CELLS."""
    feats = np.asarray(features, dtype=np.float64)
    out = np.zeros((len(seeds), feats.shape[1]))
    for i, v in enumerate(seeds):
        s, t = rowptr[v], rowptr[v + 1]
        for e in range(s, t):
            c = col[e]
            sid, did = cell_id[c], cell_id[v]
            idx = sid if (sid >= 0 and did < 0) else did if (did >= 0 and sid < 0) else n_genes if (did >= 0 and sid >= 0) else n_genes + 1
            out[i] += alpha[idx] * val[e] * feats[c]
        out[i] /= max(t - s, 1)
    return out
