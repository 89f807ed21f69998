"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np
import torch

from gnnome_assembly_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# The parity bar (BASELINE.json north_star: "within 1e-4 rel fp32"; SURVEY.md section 7
# "parity metric definition"): allclose(rtol=1e-4, atol=1e-5) AND norm-relative <= 1e-4.
RTOL = 1e-4
ATOL = 1e-5


def load_case(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    H, L, seed = int(z["H"]), int(z["L"]), int(z["seed"])
    sd = synth.synth_state_dict(H, L, seed)
    return z, sd, H, L, bool(z["batch_norm"])


def grad_stride_of(z, H):
    """Sampling stride of a fixture's grad/* and adam/* arrays (make_golden.py: every grad_stride-th element for the large models)."""
    sampled = H == 128 or ("grad_sampled" in z.files and bool(z["grad_sampled"]))
    return int(z["grad_stride"]) if sampled else 1


def sd_to_torch(sd, dtype=torch.float32, requires_grad=False):
    return {k: torch.from_numpy(np.asarray(v)).to(dtype).clone().requires_grad_(requires_grad)
            for k, v in sd.items()}


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_parity(got, want, what, rtol=RTOL, atol=ATOL, l2=1e-4):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.all(np.isfinite(got)), f"{what}: non-finite values"
    r = rel_l2(got, want)
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert r <= l2 and not bad.any(), (
        f"{what}: rel_l2={r:.3e} max_abs={np.abs(got - want).max():.3e} "
        f"violations={int(bad.sum())}/{bad.size}")


# -----------------------------------------------------------------------------------------
# gradient checks: which clause let a tensor pass (the table under profiles/), branch-exact comparison
# -----------------------------------------------------------------------------------------
GRAD_CLAUSES = {}       # pytest node id -> {clause: count}


def tally_clause(clause, count=1, forgiven=False):
    """Count a gradient tensor under the clause that decided it, for the running test.  `forgiven`: a tensor first
    counted as "miss" that the branch-exact comparison accepted -- moved from "miss" to "branch_exact"."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" ")[0]
    d = GRAD_CLAUSES.setdefault(test, {})
    d[clause] = d.get(clause, 0) + count
    if forgiven:
        d["miss"] = d.get("miss", 0) - count


def device_masks(ms, sd, e_raw_np, idx):
    """The relu branch decisions the device kernels took, reproduced exactly: every kernel tests the
    sign of a single fmaf (or of a stored pre-activation), and the sign of round(a*b+c) equals the
    sign of a*b+c evaluated in fp64 (a*b is exact there)."""
    perm = idx["perm"].long().cpu()
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    nrank = idx["nrank"].long().cpu() if "nrank" in idx else None      # internal node numbering -> the caller's
    u, w = [], []

    def branches(s):
        if getattr(s, "chunks", None):        # a layer wider than 256: one saved state per 256-column problem (engine.WIDE_CHUNK)
            uw = [branches(c) for c in s.chunks]
            return torch.cat([a for a, _ in uw], 1), torch.cat([b for _, b in uw], 1)
        uu = s.t.double() * s.stat_e[2].double() + s.stat_e[3].double()
        ww = s.z.double() * s.stat_h[2].double() + s.stat_h[3].double()
        return (uu > 0).cpu(), (ww > 0).cpu()
    for s in ms.layers:
        um, wm = branches(s)
        u.append(um[inv])                             # internal order -> edge-id order
        w.append(wm if nrank is None else wm[nrank])
    hid = (ms.pred.hid > 0).cpu()[inv]
    # encoder: ap = fmaf(w1a, x0, fmaf(w1b, x1, b))  (gnm_encoder.hip) -- inner fma rounded to fp32
    W1, b1 = sd["linear1_edge.weight"].astype(np.float64), sd["linear1_edge.bias"].astype(np.float64)
    x = e_raw_np.astype(np.float64)
    inner = (x[:, 1:2] * W1[None, :, 1] + b1[None, :]).astype(np.float32).astype(np.float64)
    a1 = torch.from_numpy((x[:, 0:1] * W1[None, :, 0] + inner) > 0)
    return {"u": u, "w": w, "hid": hid, "a1": a1}


def branch_exact_rows(src, dst, n, e_raw, pe, y, pw, sd, L, dev):
    """rows (name, rel_l2, max_abs, ref_norm) of the HIP gradients against the fp64 oracle backward evaluated
    on the SAME relu branches the device took, and the largest reference gradient norm."""
    from gnnome_assembly_amd import AssemblyGraph, engine, layers, models
    from oracle import gatedgcn_oracle as orc
    g = AssemblyGraph(src, dst, n).to(dev)
    P = {k: v.to(dev) for k, v in sd_to_torch(sd).items()}
    H = P["linear_pe.weight"].shape[0]
    Hp = layers.padded_width(H)
    if Hp != H:         # a width between the kernel instantiations: dead channels, exactly as GraphGatedGCNModel.forward pads
        P = {k: models._pad_param(k, v, H, Hp).contiguous() for k, v in P.items()}
    scores, ms = engine.model_forward(g, torch.from_numpy(e_raw).to(dev), torch.from_numpy(pe).to(dev), P, L, True)
    masks = device_masks(ms, sd, e_raw, g.index())
    loss, gs = engine.bce_with_logits(scores, torch.from_numpy(y).to(dev), pw)
    Gd = engine.model_backward(g, P, L, ms, gs)
    torch.cuda.synchronize()
    if Hp != H:
        masks["u"] = [m[:, :H] for m in masks["u"]]
        masks["w"] = [m[:, :H] for m in masks["w"]]
        for k, v in sd.items():
            if k == "predictor.W1.weight":
                Gd[k] = Gd[k].reshape(v.shape[0], 3, Hp)[:, :, :H].reshape(v.shape[0], 3 * H)
            else:
                Gd[k] = Gd[k][tuple(slice(0, d) for d in v.shape)]
    with torch.no_grad():
        _, l64, g64 = orc.manual_forward_backward(sd_to_torch(sd, torch.float64), torch.from_numpy(src), torch.from_numpy(dst),
                                                  n, torch.from_numpy(e_raw).double(), torch.from_numpy(pe).double(),
                                                  torch.from_numpy(y).double(), pw, masks=masks)
    assert abs(loss.item() - l64.item()) < 1e-5
    rows = []
    gmax = max(float(v.norm()) for v in g64.values())
    for k in g64:
        got, want = Gd[k].detach().cpu().double().numpy(), g64[k].detach().cpu().double().numpy()
        assert got.shape == want.shape, f"{k}: {got.shape} vs {want.shape}"
        rows.append((k, rel_l2(got, want), float(np.abs(got - want).max()), float(np.linalg.norm(want))))
    return rows, gmax




# -----------------------------------------------------------------------------------------
# stand-alone GatedGCN_1d variants (residual=False, in_channels != out_channels): inputs and parameters are drawn
# from seeded numpy streams here -- shared by tests/golden/make_golden_layer.py (which runs the REFERENCE layer on
# them) and by the tests -- so the fixture only stores the reference's outputs
# -----------------------------------------------------------------------------------------
LAYER_VARIANTS = {                # name: (in_channels, out_channels, batch_norm, residual argument)
    "in48_out32_bn": (48, 32, True, True),
    "in32_out32_bn_nores": (32, 32, True, False),
    "in128_out128_bn_nores": (128, 128, True, False),
    "in16_out32_ln": (16, 32, False, True),
    "in32_out32_ln_nores": (32, 32, False, False),
}


def layer_variant_case(name):
    """graph (src, dst, n), fp32 inputs h0 [N,in], e0 [E,in], functional weights wh [N,out], we [E,out] and the layer's
    state_dict (fp32 numpy, the reference's keys) for one variant."""
    cin, cout, bn, res = LAYER_VARIANTS[name]
    src, dst, n = synth.make_graph(60, seed=9, permute_edge_ids=True)
    rng = np.random.default_rng(sum(map(ord, name)))
    E = src.size
    f32 = lambda a: a.astype(np.float32)  # noqa: E731
    case = dict(src=src, dst=dst, n=n, cin=cin, cout=cout, bn=bn, res=res,
                h0=f32(rng.standard_normal((n, cin))), e0=f32(rng.standard_normal((E, cin))),
                wh=f32(rng.standard_normal((n, cout))), we=f32(rng.standard_normal((E, cout))))
    sd = {}
    b = 1.0 / np.sqrt(cin)
    for k in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3"):
        sd[k + ".weight"] = f32(rng.uniform(-b, b, size=(cout, cin)))
        sd[k + ".bias"] = f32(rng.uniform(-b, b, size=(cout,)))
    for k in ("bn_h", "bn_e"):
        sd[k + ".weight"] = f32(rng.uniform(0.5, 1.5, size=cout))
        sd[k + ".bias"] = f32(rng.uniform(-0.3, 0.3, size=cout))
    case["sd"] = sd
    return case
