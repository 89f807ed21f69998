"""CPU oracle: a plain-torch restatement of the reference's GatedGCN edge-logit path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it, and there only as the checker / the
timed CPU baseline.  The product path (gnnome_assembly_amd) never imports it and fails
loudly when its HIP library is missing.

Pinning: the reference has NO tests or golden vectors of its own
and its message-passing arithmetic lives in DGL, which is absent from /root/reference
and not installable here (requirements.txt:6-7 pins dgl-cu111==0.7.1, commented out).
The oracle is therefore pinned against outputs of the reference's own layers/ and
models/ code, imported unmodified from /root/reference in the build container with a
test-only stand-in for the five DGL builtins (tests/golden/make_golden.py); those
outputs are committed under tests/golden/*.npz and tests/test_oracle_golden.py checks
this file against them.  At the DGL boundary itself parity is UNPINNED (only DGL's
documented builtin semantics are restated).

What is restated (file:line in /root/reference):
  models/full_graph.py:12-29        encoders -> GNN stack -> predictor
  layers/processor.py:8-20          L x GatedGCN_1d
  layers/gated_gcn_full.py:35-59    parameters; :99-157 forward (the dead UDFs :61-97 are ignored)
  layers/score_predictor.py:5-25    per-edge MLP on cat(x[src], x[dst], e)
  train.py:210-211,253-255          BCEWithLogitsLoss(pos_weight), mean over edges
The "backward-message" gate on the reversed graph (gated_gcn_full.py:133-140) is the same
per-edge sum as the forward gate (B2h[dst]+B1h[src]+B3e), so one gate is computed and
aggregated twice: by destination with A2h[src] and by source with A3h[dst].

Two forms are provided:
  * model_forward / bce_loss        differentiable torch (autograd gives reference grads)
  * manual_forward_backward         hand-derived backward in the exact decomposition the
                                    HIP kernels use; checked
                                    against autograd in tests/test_oracle_golden.py
"""
from __future__ import annotations

import torch

EPS_BN = 1e-5      # nn.BatchNorm1d / nn.LayerNorm default eps (gated_gcn_full.py:55-59)
EPS_DEN = 1e-6     # gated_gcn_full.py:130,143

_LIN = ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3")


def num_layers_of(sd) -> int:
    n = 0
    while f"gnn.convs.{n}.A_1.weight" in sd:
        n += 1
    return n


def _norm(x, w, b, batch_norm: bool):
    """BatchNorm1d(track_running_stats=False) -> batch stats, biased var; or LayerNorm."""
    if batch_norm:
        mu = x.mean(0, keepdim=True)
        var = x.var(0, unbiased=False, keepdim=True)
    else:
        mu = x.mean(1, keepdim=True)
        var = x.var(1, unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + EPS_BN) * w + b


def layer_forward(sd, i, src, dst, n, h, e, batch_norm=True, residual=True):
    """One GatedGCN_1d.forward (gated_gcn_full.py:99-157). src/dst int64, edge-id order."""
    p = f"gnn.convs.{i}." if i is not None else ""
    lin = {k: (sd[p + k + ".weight"], sd[p + k + ".bias"]) for k in _LIN}
    A1h, A2h, A3h, B1h, B2h = (h @ lin[k][0].t() + lin[k][1] for k in _LIN[:5])
    B3e = e @ lin["B_3"][0].t() + lin["B_3"][1]
    t = B1h.index_select(0, src) + B2h.index_select(0, dst) + B3e          # :120-121
    u = _norm(t, sd[p + "bn_e.weight"], sd[p + "bn_e.bias"], batch_norm)   # :122
    e_out = torch.relu(u)                                                  # :123
    if residual:
        e_out = e_out + e                                                  # :124-125
    sig = torch.sigmoid(e_out)                                             # :127
    z0 = torch.zeros((n, A2h.shape[1]), dtype=h.dtype)                     # out_channels wide (in != out: :41-42)
    f_num = z0.index_add(0, dst, sig * A2h.index_select(0, src))           # :128
    f_den = z0.index_add(0, dst, sig)                                      # :129
    b_num = z0.index_add(0, src, sig * A3h.index_select(0, dst))           # :141
    b_den = z0.index_add(0, src, sig)                                      # :142
    z = A1h + f_num / (f_den + EPS_DEN) + b_num / (b_den + EPS_DEN)        # :130,143,145
    w = _norm(z, sd[p + "bn_h.weight"], sd[p + "bn_h.bias"], batch_norm)   # :147
    h_out = torch.relu(w)                                                  # :149
    if residual:
        h_out = h_out + h                                                  # :151-152
    return h_out, e_out                                                    # dropout p=0 (:154)


def predictor_forward(sd, src, dst, x, e, prefix="predictor."):
    """ScorePredictor.forward (score_predictor.py:12-25)."""
    W1, b1 = sd[prefix + "W1.weight"], sd[prefix + "W1.bias"]
    W2, b2 = sd[prefix + "W2.weight"], sd[prefix + "W2.bias"]
    data = torch.cat((x.index_select(0, src), x.index_select(0, dst), e), dim=1)
    return torch.relu(data @ W1.t() + b1) @ W2.t() + b2


def model_forward(sd, src, dst, n, e_raw, pe, batch_norm=True, return_layers=False):
    """GraphGatedGCNModel.forward (full_graph.py:22-29); the x argument is dead (:23)."""
    src = src.long()
    dst = dst.long()
    h = pe @ sd["linear_pe.weight"].t() + sd["linear_pe.bias"]
    e = torch.relu(e_raw @ sd["linear1_edge.weight"].t() + sd["linear1_edge.bias"])
    e = e @ sd["linear2_edge.weight"].t() + sd["linear2_edge.bias"]
    layers = []
    for i in range(num_layers_of(sd)):
        h, e = layer_forward(sd, i, src, dst, n, h, e, batch_norm)
        if return_layers:
            layers.append((h, e))
    scores = predictor_forward(sd, src, dst, h, e)
    return (scores, layers) if return_layers else scores


def bce_loss(scores, y, pos_weight):
    """BCEWithLogitsLoss(pos_weight=[pw]) mean-reduced (train.py:210-211,253-255)."""
    x = scores.reshape(-1)
    sp = torch.nn.functional.softplus
    return (pos_weight * y * sp(-x) + (1.0 - y) * sp(x)).mean()


def init_state_dict(node_features, edge_features, hidden_features, hidden_edge_features,
                    num_layers, hidden_edge_scores, batch_norm, nb_pos_enc, seed=0,
                    dtype=torch.float32):
    """Parameters created in the reference's order with torch default init
    (full_graph.py:12-20, processor.py:9-13, gated_gcn_full.py:35-59,
    score_predictor.py:6-10), so that the same seed gives the same weights."""
    import torch.nn as nn
    torch.manual_seed(seed)
    sd = {}

    def lin(name, i, o):
        m = nn.Linear(i, o)
        sd[name + ".weight"], sd[name + ".bias"] = m.weight.detach(), m.bias.detach()

    H = hidden_features
    lin("linear_pe", nb_pos_enc + 2, H)
    lin("linear1_edge", edge_features, hidden_edge_features)
    lin("linear2_edge", hidden_edge_features, H)
    for i in range(num_layers):
        for k in _LIN:
            lin(f"gnn.convs.{i}.{k}", H, H)
        for k in ("bn_h", "bn_e"):
            sd[f"gnn.convs.{i}.{k}.weight"] = torch.ones(H)
            sd[f"gnn.convs.{i}.{k}.bias"] = torch.zeros(H)
    lin("predictor.W1", 3 * H, hidden_edge_scores)
    lin("predictor.W2", hidden_edge_scores, 1)
    return {k: v.to(dtype).clone() for k, v in sd.items()}


# ----------------------------------------------------------------------------------------
# Hand-derived backward, in the decomposition the HIP kernels use.
# ----------------------------------------------------------------------------------------

def _seg_sum(idx, val, n):
    return torch.zeros((n, val.shape[1]), dtype=val.dtype).index_add(0, idx, val)


def _bn_fwd(x, w, b):
    mu = x.mean(0)
    var = x.var(0, unbiased=False)
    rstd = torch.rsqrt(var + EPS_BN)
    xh = (x - mu) * rstd
    return xh * w + b, xh, rstd


def _bn_bwd(gy, xh, w, rstd):
    """BNbwd(gy) = w*rstd*(gy - mean(gy) - xh*mean(gy*xh)); ggamma = sum gy*xh; gbeta = sum gy."""
    gb = gy.sum(0)
    gg = (gy * xh).sum(0)
    m = gy.shape[0]
    gx = w * rstd * (gy - gb / m - xh * (gg / m))
    return gx, gg, gb


def manual_forward_backward(sd, src, dst, n, e_raw, pe, y, pos_weight, keep=False, masks=None):
    """Forward + hand-derived backward (BatchNorm mode).  Returns (scores, loss, grads[, dbg])
    with grads keyed like the state_dict.  Mirrors SURVEY.md section 8a row 8.

    `masks` (optional) overrides the relu branch decisions of the backward pass:
    {"u": [L x bool[E,H]], "w": [L x bool[N,H]], "hid": bool[E,HS], "a1": bool[E,Q]} in edge-id /
    node order.  The network is piecewise linear in those branches; an fp32 evaluation may land on
    the other side of a kink for pre-activations within rounding distance of zero, and then the
    exact gradient OF THE BRANCH TAKEN is the meaningful reference (tests pass the device's own
    decisions here)."""
    src = src.long()
    dst = dst.long()
    L = num_layers_of(sd)
    E = src.numel()
    g = {}
    dbg = {}
    # ---- forward, keeping what the kernels keep ----
    h = pe @ sd["linear_pe.weight"].t() + sd["linear_pe.bias"]
    a1_pre = e_raw @ sd["linear1_edge.weight"].t() + sd["linear1_edge.bias"]
    a1 = torch.relu(a1_pre)
    e = a1 @ sd["linear2_edge.weight"].t() + sd["linear2_edge.bias"]
    saved = []
    for i in range(L):
        p = f"gnn.convs.{i}."
        W5 = torch.cat([sd[p + k + ".weight"] for k in _LIN[:5]], 0)
        b5 = torch.cat([sd[p + k + ".bias"] for k in _LIN[:5]], 0)
        P = h @ W5.t() + b5
        A1h, A2h, A3h, B1h, B2h = P.chunk(5, dim=1)
        t = B1h[src] + B2h[dst] + e @ sd[p + "B_3.weight"].t() + sd[p + "B_3.bias"]
        u, th, rstd_e = _bn_fwd(t, sd[p + "bn_e.weight"], sd[p + "bn_e.bias"])
        e_out = torch.relu(u) + e
        sig = torch.sigmoid(e_out)
        inv_f = 1.0 / (_seg_sum(dst, sig, n) + EPS_DEN)
        inv_b = 1.0 / (_seg_sum(src, sig, n) + EPS_DEN)
        hf = _seg_sum(dst, sig * A2h[src], n) * inv_f
        hb = _seg_sum(src, sig * A3h[dst], n) * inv_b
        z = A1h + hf + hb
        w, zh, rstd_h = _bn_fwd(z, sd[p + "bn_h.weight"], sd[p + "bn_h.bias"])
        h_out = torch.relu(w) + h
        saved.append(dict(h=h, e=e, W5=W5, A2h=A2h, A3h=A3h, u=u, th=th, rstd_e=rstd_e, sig=sig,
                          inv_f=inv_f, inv_b=inv_b, hf=hf, hb=hb, w=w, zh=zh, rstd_h=rstd_h,
                          t=t, z=z, P=P, e_out=e_out, h_out=h_out))
        h, e = h_out, e_out
    W1 = sd["predictor.W1.weight"]
    H = h.shape[1]
    W1s, W1d, W1e = W1[:, :H], W1[:, H:2 * H], W1[:, 2 * H:]
    hid = (h @ W1s.t())[src] + (h @ W1d.t())[dst] + e @ W1e.t() + sd["predictor.W1.bias"]
    r = torch.relu(hid)
    scores = r @ sd["predictor.W2.weight"].t() + sd["predictor.W2.bias"]
    loss = bce_loss(scores, y, pos_weight)
    # ---- backward ----
    x = scores.reshape(-1)
    pr = torch.sigmoid(x)
    gs = ((-pos_weight * y * (1.0 - pr) + (1.0 - y) * pr) / E).reshape(-1, 1)   # dloss/dlogit
    g["predictor.W2.weight"] = gs.t() @ r
    g["predictor.W2.bias"] = gs.sum(0)
    ghid = (gs @ sd["predictor.W2.weight"]) * (masks["hid"] if masks else (hid > 0))
    g["predictor.W1.bias"] = ghid.sum(0)
    gPs = _seg_sum(src, ghid, n)
    gPd = _seg_sum(dst, ghid, n)
    g["predictor.W1.weight"] = torch.cat([gPs.t() @ h, gPd.t() @ h, ghid.t() @ e], 1)
    gh = gPs @ W1s + gPd @ W1d
    ge = ghid @ W1e
    for i in reversed(range(L)):
        p = f"gnn.convs.{i}."
        s = saved[i]
        gw = gh * (masks["w"][i] if masks else (s["w"] > 0))
        gz, g[p + "bn_h.weight"], g[p + "bn_h.bias"] = _bn_bwd(gw, s["zh"], sd[p + "bn_h.weight"], s["rstd_h"])
        Qf = gz * s["inv_f"]
        Rf = Qf * s["hf"]
        Qb = gz * s["inv_b"]
        Rb = Qb * s["hb"]
        sig = s["sig"]
        gsig = Qf[dst] * s["A2h"][src] - Rf[dst] + Qb[src] * s["A3h"][dst] - Rb[src]
        gA2h = _seg_sum(src, sig * Qf[dst], n)
        gA3h = _seg_sum(dst, sig * Qb[src], n)
        ge_tot = ge + gsig * sig * (1.0 - sig)
        gu = ge_tot * (masks["u"][i] if masks else (s["u"] > 0))
        gt, g[p + "bn_e.weight"], g[p + "bn_e.bias"] = _bn_bwd(gu, s["th"], sd[p + "bn_e.weight"], s["rstd_e"])
        gB1h = _seg_sum(src, gt, n)
        gB2h = _seg_sum(dst, gt, n)
        g[p + "B_3.weight"] = gt.t() @ s["e"]
        g[p + "B_3.bias"] = gt.sum(0)
        ge_in = ge_tot + gt @ sd[p + "B_3.weight"]
        gP = torch.cat([gz, gA2h, gA3h, gB1h, gB2h], 1)
        gW5 = gP.t() @ s["h"]
        gb5 = gP.sum(0)
        for j, k in enumerate(_LIN[:5]):
            g[p + k + ".weight"] = gW5[j * H:(j + 1) * H]
            g[p + k + ".bias"] = gb5[j * H:(j + 1) * H]
        gh_in = gh + gP @ s["W5"]
        if keep:
            dbg[i] = dict(gz=gz, Q=torch.cat([Qf, Rf, Qb, Rb], 1), ge_tot=ge_tot, gt=gt, gP=gP,
                          ge_in=ge_in, gh_in=gh_in, gh_out=gh, ge_out=ge, **s)
        gh, ge = gh_in, ge_in
    g["linear_pe.weight"] = gh.t() @ pe
    g["linear_pe.bias"] = gh.sum(0)
    g["linear2_edge.weight"] = ge.t() @ a1
    g["linear2_edge.bias"] = ge.sum(0)
    ga1 = (ge @ sd["linear2_edge.weight"]) * (masks["a1"] if masks else (a1_pre > 0))
    g["linear1_edge.weight"] = ga1.t() @ e_raw
    g["linear1_edge.bias"] = ga1.sum(0)
    if keep:
        dbg["gs"] = gs
        dbg["ghid"] = ghid
        dbg["hid"] = hid
        return scores, loss, g, dbg
    return scores, loss, g


def adam_step(params, grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """First torch.optim.Adam step from zero state (train.py:209,258): p - lr*g/(|g|+eps)."""
    out = {}
    for k, p in params.items():
        gk = grads[k]
        m = (1 - betas[0]) * gk
        v = (1 - betas[1]) * gk * gk
        mhat = m / (1 - betas[0])
        vhat = v / (1 - betas[1])
        out[k] = p - lr * mhat / (vhat.sqrt() + eps)
    return out
