"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libgnm.so via ctypes), against the CPU oracle on the same seeded inputs and against the
committed golden vectors generated from the reference's own code.

Tolerance (BASELINE.json north_star "within 1e-4 rel fp32"; SURVEY.md section 7 parity metric):
allclose(rtol=1e-4, atol=1e-5) and ||d||2/||ref||2 <= 1e-4 on edge logits; the same bar on loss
and on every parameter gradient measured norm-relative per tensor (rel_l2 <= 1e-3 for
gradients whose reference norm is itself round-off, see GRAD_* below)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files
from helpers import GOLDEN, load_case, grad_stride_of, sd_to_torch, rel_l2, assert_parity, tally_clause, RTOL, ATOL

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f16x2", "bf16x3", "f32"])
def matmul_mode(request):
    """Every test of this file runs under ALL THREE matmul modes of the fused kernels (include/gnm.h):
    "f16x2" -- the library default, the mode bench.py's `value` is measured in: two fp16 terms of a power-of-two
    multiple, three MFMAs per product --, "bf16x3" (the exact three-term split, six MFMAs
    per product; the default of rounds 2-4) and the fp32-MFMA mode.  Tests that never reach a fused kernel are marked
    `mode_independent` and run once."""
    from gnnome_assembly_amd import _lib
    if request.param != _lib.DEFAULT_MATMUL_MODE and (request.node.get_closest_marker("mode_independent")
                                                      or request.node.get_closest_marker("default_mode_only")):
        pytest.skip("runs once (does not depend on the matmul mode, or too large to run twice)")
    _lib.set_matmul_mode(request.param)
    yield request.param
    _lib.set_matmul_mode(_lib.DEFAULT_MATMUL_MODE)

GRAD_L2 = 2e-4          # norm-relative bar for one parameter-gradient tensor (fp32 vs fp64 oracle)
GRAD_ABS_FLOOR = 2e-7   # gradients that are analytically zero (biases in front of a BatchNorm)
# Gradients that pass through BatchNorm_e backward (B_1/B_2/B_3, bn_e) are differences of large sums and flip with single
# relu-boundary elements: the reference's OWN fp32 arithmetic (the oracle run in fp32) differs from fp64 by up to ~1e-3
# norm-relative on them.  Rounds 1-4 let such a tensor pass when it was no further from the fp64 oracle than NOISE_X times the
# fp32 oracle is -- a bar that moves with the fixture and bounded nothing.  Since round 5 a BatchNorm model has NO
# noise clause: a tensor outside GRAD_L2 must be EXACT (rel-L2 <= BRANCH_L2 = 5e-5) against the fp64 backward evaluated on the
# relu branches the device took -- the network is piecewise linear in those branches, so that comparison has no kink
# ambiguity and no fixture-dependent slack.  The noise clause survives only where there is no branch-exact oracle (LayerNorm
# models: none of their tensors has ever needed it).
NOISE_X = 3.0


def _grad_ok(r_ours, max_abs, floor, r_ref32=None):
    """Which clause decides this tensor is tallied per test (helpers.GRAD_CLAUSES -> gpurun_out/grad_clauses.json, committed
    under profiles/).  BatchNorm models (r_ref32 is None): rel-L2 <= GRAD_L2 or "miss" -- the caller takes the misses to the
    branch-exact comparison (_branch_exact_or_fail), where the absolute floor is the last resort.  LayerNorm models (no
    branch-exact oracle): rel-L2, then within NOISE_X of the fp32 oracle's own distance from the fp64 one, then the floor.
    tests/test_zz_grad_clause_budget.py fails the suite when a test takes more "noise" / "floor" escapes than the committed
    baseline."""
    if r_ref32 is None:
        clause = "l2" if r_ours <= GRAD_L2 else "miss"
    else:
        clause = ("l2" if r_ours <= GRAD_L2 else "noise" if r_ours <= NOISE_X * r_ref32 + 1e-6 else
                  "floor" if max_abs <= floor else "miss")
    tally_clause(clause)
    return clause != "miss"


def _branch_exact_or_fail(bad, exact, bgmax, what, floor=None):
    """bad: rows (name, ...) that missed the plain bar; exact: name -> (name, rel_l2, max_abs, ref_norm) against the fp64
    backward on the device's branches.  Each must be exact (rel-L2 <= BRANCH_L2: tallied "branch_exact") or, failing that, under
    the absolute floor (max_abs <= max(GRAD_ABS_FLOOR, 1e-6 x the largest gradient norm): "floor" -- the gradients that are
    analytically zero, e.g. of a bias in front of a BatchNorm)."""
    floor = max(GRAD_ABS_FLOOR, 1e-6 * bgmax) if floor is None else floor
    ex = [b for b in bad if exact[b[0]][1] <= BRANCH_L2]
    fl = [b for b in bad if exact[b[0]][1] > BRANCH_L2 and exact[b[0]][2] <= floor]
    still = [b for b in bad if exact[b[0]][1] > BRANCH_L2 and exact[b[0]][2] > floor]
    tally_clause("branch_exact", len(ex), forgiven=True)
    tally_clause("floor", len(fl), forgiven=True)
    assert not still, (f"{what}: gradient tensors outside rel-L2 {GRAD_L2:g} of the fp64 oracle AND not exact ({BRANCH_L2:g}) for the "
                       f"relu branches the device took: {[(b, exact[b[0]]) for b in still]}")


def _oracle_grads(z, sd, dtype, batch_norm=True):
    from oracle import gatedgcn_oracle as orc
    p = sd_to_torch(sd, dtype, requires_grad=True)
    s = orc.model_forward(p, torch.from_numpy(z["src"]), torch.from_numpy(z["dst"]), int(z["n"]),
                          torch.from_numpy(z["e_raw"]).to(dtype), torch.from_numpy(z["pe"]).to(dtype), batch_norm)
    orc.bce_loss(s, torch.from_numpy(z["y"]).to(dtype), float(z["pos_weight"])).backward()
    return {k: v.grad.double().numpy() for k, v in p.items()}


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _report(rows, path=None):
    txt = "\n".join(f"{name:28s} rel_l2={r:.3e} max_abs={m:.3e} ref_norm={n:.3e}" for name, r, m, n in rows)
    print(txt)
    os.makedirs("gpurun_out", exist_ok=True)
    if path:
        with open(os.path.join("gpurun_out", path), "w") as f:
            f.write(txt + "\n")
    return txt


def _cmp(name, got, want, rows):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = want.detach().cpu().double().numpy() if torch.is_tensor(want) else np.asarray(want, np.float64)
    assert got.shape == want.shape, f"{name}: {got.shape} vs {want.shape}"
    rows.append((name, rel_l2(got, want), float(np.abs(got - want).max()), float(np.linalg.norm(want))))


# -----------------------------------------------------------------------------------------
# library / GEMM
# -----------------------------------------------------------------------------------------

@pytest.mark.mode_independent
def test_library_loaded_and_device():
    from gnnome_assembly_amd import _lib
    lib = _lib.load()
    assert lib.gnm_abi_version() == _lib.ABI_VERSION == 7
    assert lib.gnm_num_cus() >= 64
    print("CUs:", lib.gnm_num_cus(), torch.cuda.get_device_name(0))


@pytest.mark.parametrize("mode,M,N,K", [
    (0, 256, 128, 128), (0, 1000, 640, 128), (0, 777, 64, 128), (0, 300, 128, 18), (0, 513, 16, 2),
    (1, 256, 128, 128), (1, 1000, 128, 640), (1, 333, 16, 128), (1, 500, 128, 64),
    (2, 128, 128, 4096), (2, 640, 128, 5000), (2, 64, 128, 3001), (2, 128, 18, 1000), (2, 16, 2, 777),
    (2, 128, 128, 100000),
    # big-M shapes with 128-multiple other dimensions: the split-mode route (gemm_rows_b3_k / tn_tr_k classes) under
    # bf16x3, the fp32-MFMA kernel under f32 -- the shapes of a hidden-256 model (the reference's default width)
    (0, 5000, 256, 256), (0, 3001, 1280, 256), (0, 2048, 128, 128), (1, 4100, 256, 256), (1, 2500, 256, 1280),
    (2, 256, 256, 9000), (2, 1280, 256, 5003), (2, 256, 128, 4096), (2, 128, 18, 9001),
])
def test_gemm_f32(mode, M, N, K):
    from gnnome_assembly_amd import engine
    dev = _dev()
    rng = np.random.default_rng(mode * 1000 + M + N + K)
    shapes = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[mode]
    A = rng.standard_normal(shapes[0]).astype(np.float32)
    B = (rng.standard_normal(shapes[1]) + 0.25).astype(np.float32)   # asymmetric
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    ref = (A64 @ B64.T) if mode == 0 else (A64 @ B64) if mode == 1 else (A64.T @ B64)
    dA, dB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    engine.gemm(mode, dA, dB, out)
    scale = np.abs(A64).sum() / A.size * np.abs(B64).sum() / B.size * K
    err = np.abs(out.cpu().double().numpy() - ref).max()
    assert err <= 2e-6 * max(scale, 1.0) * max(1.0, np.sqrt(K) / 8), f"plain: max err {err:.3e} (scale {scale:.3e})"
    # epilogue: bias + resid + relu, strided output view
    big = torch.zeros((M, N + 8), device=dev)
    outv = big[:, 4:4 + N]
    engine.gemm(mode, dA, dB, outv, bias=torch.from_numpy(bias).to(dev), resid=torch.from_numpy(R).to(dev), relu=True)
    ref2 = np.maximum(ref + bias + R, 0.0)
    err2 = np.abs(outv.cpu().double().numpy() - ref2).max()
    assert err2 <= 2e-6 * max(scale, 1.0) * max(1.0, np.sqrt(K) / 8) + 1e-6, f"epilogue: max err {err2:.3e}"
    assert float(big[:, :4].abs().max()) == 0.0 and float(big[:, 4 + N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(256, 256, 9000), (1280, 256, 5003), (128, 128, 4096), (64, 128, 3001), (128, 18, 1000),
                                   (128, 18, 100003), (128, 32, 4500), (128, 7, 5003), (128, 16, 4096)])   # skinny fp32-MFMA route
def test_gemm_tn_colsum(M, N, K):
    """gnm_gemm_tn_colsum: a Linear's weight and bias gradient in one call (split route where it applies, gemm + colsum
    otherwise) against fp64."""
    from gnnome_assembly_amd import engine
    dev = _dev()
    rng = np.random.default_rng(M + N + K)
    A = (rng.standard_normal((K, M)) * np.exp(rng.standard_normal((K, 1)))).astype(np.float32)   # rows of mixed magnitude
    B = (rng.standard_normal((K, N)) + 0.25).astype(np.float32)
    big = torch.full((M, N + 4), float("nan"), device=dev)
    C_ = big[:, :N]
    cs = engine.gemm_tn_colsum(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), C_)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    assert rel_l2(C_.cpu().double().numpy(), ref) <= 2e-6
    assert rel_l2(cs.cpu().double().numpy(), A.astype(np.float64).sum(0)) <= 2e-6
    assert bool(torch.isnan(big[:, N:]).all())


# -----------------------------------------------------------------------------------------
# one layer, kernel by kernel, against the oracle's hand-derived decomposition
# -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("fname", ["small_h64l1_s1.npz", "tiny_h64l1_s0.npz", "small_h128l8_s1.npz"])
def test_layer_kernels_vs_oracle(fname):
    from gnnome_assembly_amd import AssemblyGraph, engine
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    z, sd, H, L, bn = load_case(fname)
    src, dst, n = z["src"], z["dst"], int(z["n"])
    p64 = sd_to_torch(sd, torch.float64)
    with torch.no_grad():
        _, _, g64, dbg = orc.manual_forward_backward(
            p64, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(z["e_raw"]).double(),
            torch.from_numpy(z["pe"]).double(), torch.from_numpy(z["y"]).double(), float(z["pos_weight"]), keep=True)
    graph = AssemblyGraph(src, dst, n, node_order="keep").to(dev)     # the engine-level calls below take node rows as they are
    idx = graph.index()
    assert "nperm" not in idx
    perm = idx["perm"].long().cpu()
    E = src.size
    li = L - 1          # check the last layer (its incoming gradients come straight from the predictor)
    d = dbg[li]
    P32 = {k: v.to(dev) for k, v in sd_to_torch(sd).items()}
    prm = engine.layer_params(P32, li)
    f = lambda t: t.float().contiguous().to(dev)  # noqa: E731
    h_in, e_in = f(d["h"]), f(d["e"][perm])
    h_out, e_out, s = engine.layer_forward(idx, n, E, H, prm, h_in, e_in, True)
    torch.cuda.synchronize()
    rows = []
    _cmp("P", s.P, d["P"], rows)
    _cmp("t", s.t, d["t"][perm], rows)
    _cmp("stat_e.mean", s.stat_e[0], d["t"].mean(0), rows)
    _cmp("stat_e.rstd", s.stat_e[1], d["rstd_e"], rows)
    _cmp("e_out", e_out, d["e_out"][perm], rows)
    _cmp("hf", s.hf, d["hf"], rows)
    _cmp("inv_f", s.inv_f, d["inv_f"], rows)
    _cmp("hb", s.hb, d["hb"], rows)
    _cmp("inv_b", s.inv_b, d["inv_b"], rows)
    _cmp("z", s.z, d["z"], rows)
    _cmp("stat_h.rstd", s.stat_h[1], d["rstd_h"], rows)
    _cmp("h_out", h_out, d["h_out"], rows)
    nfwd = len(rows)
    gh_out, ge = f(d["gh_out"]), f(d["ge_out"][perm])
    gh_in, ge_in, g = engine.layer_backward(idx, n, E, H, prm, s, gh_out, ge)
    torch.cuda.synchronize()
    pfx = f"gnn.convs.{li}."
    _cmp("gh_in", gh_in, d["gh_in"], rows)
    _cmp("ge_in", ge_in, d["ge_in"][perm], rows)
    gW5 = torch.cat([g64[pfx + k + ".weight"] for k in engine.LIN5], 0)
    gb5 = torch.cat([g64[pfx + k + ".bias"] for k in engine.LIN5], 0)
    _cmp("gW5", g["W5"], gW5, rows)
    _cmp("gb5[A2,A3]", g["b5"][H:3 * H], gb5[H:3 * H], rows)
    _cmp("gW3", g["W3"], g64[pfx + "B_3.weight"], rows)
    _cmp("g gamma_e", g["gamma_e"], g64[pfx + "bn_e.weight"], rows)
    _cmp("g beta_e", g["beta_e"], g64[pfx + "bn_e.bias"], rows)
    _cmp("g gamma_h", g["gamma_h"], g64[pfx + "bn_h.weight"], rows)
    _cmp("g beta_h", g["beta_h"], g64[pfx + "bn_h.bias"], rows)
    _report(rows, f"layer_{fname}.txt")
    bad = [r for r in rows[:nfwd] if r[1] > 2e-5]
    assert not bad, f"forward mismatches: {bad}"
    miss = [r for r in rows[nfwd:] if not _grad_ok(r[1], r[2], 0.0)]
    if miss:
        # outside the plain bar: the same comparison against the fp64 backward evaluated on the relu branches THIS layer took
        # on the device (the other layers, the predictor and the encoder keep the oracle's own branches: the layer's incoming
        # gradients gh_out / ge_out come from above and do not depend on its branches)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel())
        ud = ((s.t.double() * s.stat_e[2].double() + s.stat_e[3].double()) > 0).cpu()[inv]
        wd = ((s.z.double() * s.stat_h[2].double() + s.stat_h[3].double()) > 0).cpu()
        a1_pre = torch.from_numpy(z["e_raw"]).double() @ p64["linear1_edge.weight"].t() + p64["linear1_edge.bias"]
        masks = {"u": [dbg[i]["u"] > 0 for i in range(L)], "w": [dbg[i]["w"] > 0 for i in range(L)], "hid": dbg["hid"] > 0,
                 "a1": a1_pre > 0}
        masks["u"][li], masks["w"][li] = ud, wd
        with torch.no_grad():
            _, _, gx, dx = orc.manual_forward_backward(
                p64, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(z["e_raw"]).double(),
                torch.from_numpy(z["pe"]).double(), torch.from_numpy(z["y"]).double(), float(z["pos_weight"]), keep=True, masks=masks)
        assert torch.equal(dx[li]["gh_out"], d["gh_out"]) and torch.equal(dx[li]["ge_out"], d["ge_out"])
        xrows = []
        _cmp("gh_in", gh_in, dx[li]["gh_in"], xrows)
        _cmp("ge_in", ge_in, dx[li]["ge_in"][perm], xrows)
        _cmp("gW5", g["W5"], torch.cat([gx[pfx + k + ".weight"] for k in engine.LIN5], 0), xrows)
        _cmp("gb5[A2,A3]", g["b5"][H:3 * H], torch.cat([gx[pfx + k + ".bias"] for k in engine.LIN5], 0)[H:3 * H], xrows)
        _cmp("gW3", g["W3"], gx[pfx + "B_3.weight"], xrows)
        _cmp("g gamma_e", g["gamma_e"], gx[pfx + "bn_e.weight"], xrows)
        _cmp("g beta_e", g["beta_e"], gx[pfx + "bn_e.bias"], xrows)
        _cmp("g gamma_h", g["gamma_h"], gx[pfx + "bn_h.weight"], xrows)
        _cmp("g beta_h", g["beta_h"], gx[pfx + "bn_h.bias"], xrows)
        _report(xrows, f"layer_branch_{fname}.txt")
        _branch_exact_or_fail(miss, {r[0]: r for r in xrows}, max(r[3] for r in xrows), f"{fname} layer {li}")
    # biases that feed a BatchNorm directly have an analytically zero gradient
    zero_b = torch.cat([g["b5"][:H], g["b5"][3 * H:], g["b3"]]).abs().max().item()
    scale = float(gW5.abs().max())
    assert zero_b <= max(GRAD_ABS_FLOOR, 1e-4 * scale), f"pre-BN bias gradients should be ~0, got {zero_b:.3e}"


# -----------------------------------------------------------------------------------------
# whole model against the golden vectors (reference outputs) and the oracle
# -----------------------------------------------------------------------------------------

def _run_model(z, sd, H, L, dev, loss_kind="fused"):
    import gnnome_assembly_amd as G
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, bool(z["batch_norm"]), 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev)
    graph = G.AssemblyGraph(z["src"], z["dst"], int(z["n"])).to(dev)
    x = torch.ones(int(z["n"]), 1, device=dev)
    e = torch.from_numpy(z["e_raw"]).to(dev)
    pe = torch.from_numpy(z["pe"]).to(dev)
    y = torch.from_numpy(z["y"]).to(dev)
    pw = float(z["pos_weight"])
    if loss_kind == "fused":
        crit = G.BCEWithLogitsLoss(pos_weight=pw)
    else:   # the reference's own criterion (train.py:210-211) on top of our logits
        crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([pw], device=dev))
    return model, graph, x, e, pe, y, crit


@pytest.mark.parametrize("fname", golden_files())
def test_model_matches_golden(fname):
    dev = _dev()
    z, sd, H, L, bn = load_case(fname)
    model, graph, x, e, pe, y, crit = _run_model(z, sd, H, L, dev)
    model.train()
    scores = model(graph, x, e, pe)
    assert scores.shape == (z["src"].size, 1)
    loss = crit(scores.squeeze(-1), y)
    loss.backward()
    torch.cuda.synchronize()
    # logits: the reference's fp64 run is the tie-breaker, its fp32 run shows fp32 noise level
    assert_parity(scores.detach().cpu().numpy(), z["scores64"], f"{fname} logits vs reference fp64")
    ref_noise = rel_l2(z["scores32"], z["scores64"])
    ours = rel_l2(scores.detach().cpu().numpy(), z["scores64"])
    print(f"{fname}: logits rel_l2 ours={ours:.2e} reference-fp32={ref_noise:.2e}")
    assert abs(loss.item() - float(z["loss64"])) <= 1e-5 * max(1.0, abs(float(z["loss64"])))
    stride = grad_stride_of(z, H)
    g32 = None if bn else _oracle_grads(z, sd, torch.float32, bn)  # LayerNorm only: the reference arithmetic in fp32 (noise level)
    rows, bad = [], []
    gmax = max(float(np.linalg.norm(z["grad/" + k])) for k, _ in model.named_parameters())
    for k, prm in model.named_parameters():
        got = prm.grad.detach().cpu().double().numpy().reshape(-1)[::stride]
        want = z["grad/" + k]                      # reference fp64 (golden)
        _cmp(k, got, want, rows)
        r32 = None if bn else rel_l2(g32[k].reshape(-1)[::stride], want)
        if not _grad_ok(rows[-1][1], rows[-1][2], max(GRAD_ABS_FLOOR, 1e-6 * gmax), r32):
            bad.append(rows[-1] + (r32,))
    _report(rows, f"model_{fname}.txt")
    if bad and bn:
        # A tensor outside the plain bar passes ONLY if the whole deviation is a relu decision that fell the other way in
        # fp32 (every fp32 evaluation, the reference's own included, flips some): against the fp64 backward evaluated on the
        # branches the device took it must agree to fp32 round-off.  No noise clause (see GRAD_L2 above).
        brows, bgmax = _branch_exact_rows(z["src"], z["dst"], int(z["n"]), z["e_raw"], z["pe"], z["y"],
                                          float(z["pos_weight"]), sd, L, dev)
        _branch_exact_or_fail(bad, {r[0]: r for r in brows}, bgmax, fname)
        bad = []
    assert not bad, f"gradient mismatches (name, rel_l2, max_abs, ref_norm, reference-fp32 rel_l2): {bad}"
    # eval mode == train mode (BatchNorm has no running stats: gated_gcn_full.py:55-56)
    model.eval()
    with torch.no_grad():
        s2 = model(graph, x, e, pe)
    assert torch.equal(s2, scores.detach())


@pytest.mark.parametrize("fname", ["tiny_h64l1_s0.npz", "small_h64l1_s1.npz"])
def test_native_64_wide_route_matches_golden(fname, monkeypatch):
    """Since round 6 the modules run a 64-wide model zero-padded to 128 on the fused kernels and sweeps (layers.RUN_WIDTHS: faster
    at every size measured).  The 64-wide kernels stay built -- GNM_NATIVE_64=1 -- and this keeps them pinned to the reference at
    model level: the same golden comparison as test_model_matches_golden with the native widths switched back on, and the two
    routes agree with each other to fp32 round-off."""
    from gnnome_assembly_amd import layers
    dev = _dev()
    z, sd, H, L, bn = load_case(fname)
    out = {}
    for tag, widths in (("padded", (32, 128, 256)), ("native", layers.KERNEL_WIDTHS)):
        monkeypatch.setattr(layers, "RUN_WIDTHS", widths)
        assert layers.padded_width(64) == (128 if tag == "padded" else 64)
        model, graph, x, e, pe, y, crit = _run_model(z, sd, H, L, dev)
        scores = model(graph, x, e, pe)
        loss = crit(scores.squeeze(-1), y)
        loss.backward()
        torch.cuda.synchronize()
        assert_parity(scores.detach().cpu().numpy(), z["scores64"], f"{fname} logits vs reference fp64 ({tag})")
        assert abs(loss.item() - float(z["loss64"])) <= 1e-5 * max(1.0, abs(float(z["loss64"])))
        out[tag] = (scores.detach().cpu().numpy(), {k: p.grad.detach().cpu().double().numpy() for k, p in model.named_parameters()})
        for k, g in out[tag][1].items():
            assert g.shape == tuple(sd[k].shape), (tag, k, g.shape)
    assert rel_l2(out["padded"][0], out["native"][0]) <= 5e-6
    gmax = max(float(np.linalg.norm(z["grad/" + k])) for k in out["native"][1])
    for k in out["native"][1]:
        a, b = out["padded"][1][k].reshape(-1), out["native"][1][k].reshape(-1)
        assert rel_l2(a, b) <= 1e-3 or np.abs(a - b).max() <= max(GRAD_ABS_FLOOR, 1e-6 * gmax), (k, rel_l2(a, b))


@pytest.mark.parametrize("fname", ["tiny_h64l1_s0.npz", "small_h64l1_s0.npz", "small_h128l8_s0.npz"])
def test_three_adam_steps_match_reference(fname):
    """train.py:252-258 loop on one graph: loss sequence of 3 Adam steps (golden loss_seq64)."""
    dev = _dev()
    z, sd, H, L, bn = load_case(fname)
    model, graph, x, e, pe, y, crit = _run_model(z, sd, H, L, dev, loss_kind="torch")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    seq = []
    for _ in range(3):
        model.train()
        pred = model(graph, x, e, pe).squeeze(-1)
        loss = crit(pred, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        seq.append(loss.item())
    ref = z["loss_seq64"]
    print(fname, "loss seq", seq, "ref", list(ref))
    assert np.abs(np.array(seq) - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    # TP/TN/FP/FN of the first step (utils.py:217-223) from a fresh model
    model2, graph, x, e, pe, y, crit = _run_model(z, sd, H, L, dev)
    with torch.no_grad():
        pred = torch.round(torch.sigmoid(model2(graph, x, e, pe).squeeze(-1)))
    tfpn = [int(((pred == a) & (y == b)).sum()) for a, b in ((1, 1), (0, 0), (1, 0), (0, 1))]
    assert sum(abs(a - int(b)) for a, b in zip(tfpn, z["tfpn"])) <= 2, (tfpn, list(z["tfpn"]))


@pytest.mark.mode_independent
def test_fused_bce_matches_torch():
    import gnnome_assembly_amd as G
    dev = _dev()
    rng = np.random.default_rng(5)
    x = torch.from_numpy((rng.standard_normal(100003) * 4).astype(np.float32)).to(dev).requires_grad_(True)
    y = torch.from_numpy((rng.random(100003) < 0.8).astype(np.float32)).to(dev)
    pw = 0.27
    l1 = G.BCEWithLogitsLoss(pw)(x, y)
    l1.backward()
    g1 = x.grad.clone()
    x.grad = None
    l2 = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([pw], device=dev))(x, y)
    l2.backward()
    assert abs(l1.item() - l2.item()) < 1e-6
    assert float((g1 - x.grad).abs().max()) < 1e-10 + 1e-6 * float(x.grad.abs().max())


# -----------------------------------------------------------------------------------------
# stand-alone modules (edge-id order at every module boundary, like DGL)
# -----------------------------------------------------------------------------------------

@pytest.mark.mode_independent
@pytest.mark.parametrize("H,width", [(32, 32), (32, 20), (64, 64), (64, 48), (128, 128), (128, 96), (256, 256), (256, 200)])
def test_layernorm_row_statistics_at_every_kernel_width(H, width):
    """The row sums of the LayerNorm kernels (gnm_ln.h row_sum: DPP adds inside a 16-lane row, v_permlane16_swap across the two rows of a
    32-lane group, one shuffle across the halves of a wave) at every kernel width and at a zero-padded real width, through
    gnm_ln_node_update_fwd against numpy in fp64.  (Round 6: the first DPP version read one half of the swap twice at 32 and 64 lanes
    per row -- a 10 % error the schedule-against-schedule tests could not see; this one does.)"""
    import ctypes as C
    from gnnome_assembly_amd import engine
    dev = _dev()
    rng = np.random.default_rng(H + width)
    N = 1003
    z = rng.standard_normal((N, H)).astype(np.float32) * np.exp(rng.uniform(-3, 3, (N, 1))).astype(np.float32)
    z[:, width:] = 0
    ga = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    be = (0.1 * rng.standard_normal(H)).astype(np.float32)
    zt, gt, bt = (torch.from_numpy(a).to(dev) for a in (z, ga, be))
    out = torch.empty(N, H, device=dev)
    engine._call("gnm_ln_node_update_fwd", N, H, engine._ptr(zt), engine._ptr(gt), engine._ptr(bt), C.c_void_p(0), engine._ptr(out), width,
                 engine._stream())
    torch.cuda.synchronize()
    zz = z[:, :width].astype(np.float64)
    ref = np.maximum((zz - zz.mean(1, keepdims=True)) / np.sqrt(zz.var(1, keepdims=True) + 1e-5) * ga[:width] + be[:width], 0)
    assert rel_l2(out.cpu().numpy()[:, :width], ref) <= 5e-7


@pytest.mark.mode_independent
def test_standalone_layer_dropout():
    """gated_gcn_full.py:154: dropout on the node output after the residual, training mode only (never enabled by the
    model).  Kept elements are the p = 0 output scaled by 1/(1-p), e is untouched, eval mode is the p = 0 output."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    dev = _dev()
    H = 64
    src, dst, n = synth.make_graph(300, seed=5)
    rng = np.random.default_rng(5)
    h0 = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32)).to(dev)
    e0 = torch.from_numpy(rng.standard_normal((src.size, H)).astype(np.float32)).to(dev)
    graph = G.AssemblyGraph(src, dst, n).to(dev)
    torch.manual_seed(0)
    plain = G.layers.GatedGCN_1d(H, H, True).to(dev)
    drop = G.layers.GatedGCN_1d(H, H, True, dropout=0.5).to(dev)
    drop.load_state_dict(plain.state_dict())
    h_ref, e_ref = plain(graph, h0, e0)
    hd, ed = drop(graph, h0.clone().requires_grad_(True), e0)
    kept = hd != 0
    frac = float(kept.float().mean())
    assert 0.4 < frac < 0.6, frac
    assert torch.equal(ed, e_ref)
    assert torch.allclose(hd[kept], 2.0 * h_ref[kept], rtol=0, atol=0)
    hd.sum().backward()                                   # the mask is part of the autograd graph
    drop.eval()
    he, _ = drop(graph, h0, e0)
    assert torch.equal(he, h_ref)


def test_standalone_modules_match_oracle():
    import gnnome_assembly_amd as G
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    z, sd, H, L, bn = load_case("small_h64l1_s1.npz")
    src, dst, n = z["src"], z["dst"], int(z["n"])
    E = src.size
    rng = np.random.default_rng(11)
    h0 = rng.standard_normal((n, H)).astype(np.float32)
    e0 = rng.standard_normal((E, H)).astype(np.float32)
    graph = G.AssemblyGraph(src, dst, n).to(dev)
    gnn = G.layers.GraphGatedGCN(1, H, True)
    pred = G.layers.ScorePredictor(H, 64)
    gnn.load_state_dict({k[len("gnn."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("gnn.")})
    pred.load_state_dict({k[len("predictor."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("predictor.")})
    gnn.to(dev), pred.to(dev)
    h = torch.from_numpy(h0).to(dev).requires_grad_(True)
    e = torch.from_numpy(e0).to(dev).requires_grad_(True)
    h1, e1 = gnn(graph, h, e)
    s = pred(graph, h1, e1)
    w = torch.from_numpy(rng.standard_normal((E, 1)).astype(np.float32)).to(dev)
    (s * w).sum().backward()
    torch.cuda.synchronize()
    # oracle, fp64 autograd
    p = sd_to_torch(sd, torch.float64, requires_grad=True)
    hh = torch.from_numpy(h0).double().requires_grad_(True)
    ee = torch.from_numpy(e0).double().requires_grad_(True)
    ts, td = torch.from_numpy(src).long(), torch.from_numpy(dst).long()
    rh, re = orc.layer_forward(p, 0, ts, td, n, hh, ee)
    rs = orc.predictor_forward(p, ts, td, rh, re)
    (rs * w.cpu().double()).sum().backward()
    rows = []
    _cmp("h1", h1, rh, rows)
    _cmp("e1 (edge-id order)", e1, re, rows)
    _cmp("scores", s, rs, rows)
    _cmp("d/dh", h.grad, hh.grad, rows)
    _cmp("d/de (edge-id order)", e.grad, ee.grad, rows)
    for k, prm in list(gnn.named_parameters()) + list(pred.named_parameters()):
        full = ("gnn." if k.startswith("convs") else "predictor.") + k
        if any(b in full for b in ("A_1.bias", "B_1.bias", "B_2.bias", "B_3.bias")):
            continue
        _cmp(full, prm.grad, p[full].grad, rows)
    _report(rows, "standalone.txt")
    bad = [r for r in rows if r[1] > GRAD_L2]
    assert not bad, bad


@pytest.mark.parametrize("bn", [True, False])
def test_odd_hidden_width_runs_zero_padded(bn):
    """nn.Linear(in, out) of the reference takes any width (gated_gcn_full.py:44-50); the row kernels are instantiated for
    32 / 64 / 128 / 256.  A width in between (96) runs on the next one up with dead channels: model logits, loss and every
    parameter gradient against the fp64 oracle at the usual bars, state_dict shapes untouched.  batch_norm=False: the
    reference's nn.LayerNorm(out_channels) (gated_gcn_full.py:58-59) takes its row statistics over the 96 (48) REAL channels --
    the LayerNorm kernels get the real width and leave the dead channels out."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    H, L, seed = 96, 2, 13
    src, dst, n = synth.make_graph(400, seed, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, seed)
    sd = synth.synth_state_dict(H, L, seed)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, bn, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    assert model.gnn.convs[0].A_1.weight.shape == (H, H)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    s = model(g, None, e, pe)
    loss = G.BCEWithLogitsLoss(float(inp["pos_weight"]))(s.squeeze(-1), y)
    loss.backward()
    torch.cuda.synchronize()
    p64 = sd_to_torch(sd, torch.float64, requires_grad=True)
    ts, td = torch.from_numpy(src), torch.from_numpy(dst)
    r = orc.model_forward(p64, ts, td, n, torch.from_numpy(inp["e"]).double(), torch.from_numpy(inp["pe"]).double(), bn)
    l64 = orc.bce_loss(r, torch.from_numpy(inp["y"]).double(), float(inp["pos_weight"]))
    l64.backward()
    assert_parity(s.detach().cpu().numpy(), r.detach().numpy(), "H=96 logits")
    assert abs(loss.item() - l64.item()) <= 1e-5 * abs(l64.item()) + 1e-7
    p32 = sd_to_torch(sd, torch.float32, requires_grad=True)
    if not bn:      # LayerNorm: the reference arithmetic in fp32 gives the noise level (there is no branch-exact oracle for it)
        orc.bce_loss(orc.model_forward(p32, ts, td, n, torch.from_numpy(inp["e"]), torch.from_numpy(inp["pe"]), bn),
                     torch.from_numpy(inp["y"]), float(inp["pos_weight"])).backward()
    gmax = max(float(v.grad.norm()) for v in p64.values())
    bad = []
    for k, prm in model.named_parameters():
        assert prm.grad.shape == p64[k].grad.shape
        want = p64[k].grad.numpy()
        ro = rel_l2(prm.grad.cpu().numpy(), want)
        rr = None if bn else rel_l2(p32[k].grad.double().numpy(), want)
        if not _grad_ok(ro, float(np.abs(prm.grad.cpu().numpy() - want).max()), GRAD_ABS_FLOOR * max(gmax, 1.0), rr):
            bad.append((k, ro, rr))
    if bad and bn:
        brows, bgmax = _branch_exact_rows(src, dst, n, inp["e"], inp["pe"], inp["y"], float(inp["pos_weight"]), sd, L, dev)
        _branch_exact_or_fail(bad, {r[0]: r for r in brows}, bgmax, "H=96")
        bad = []
    assert not bad, bad
    # the stand-alone layer at an odd width, residual on (in == out == 48 -> padded to 64)
    lay = G.layers.GatedGCN_1d(48, 48, bn).to(dev)
    rng = np.random.default_rng(4)
    with torch.no_grad():       # norm weights / biases off their init values: a dead channel must not see them either
        for nm in ("bn_h", "bn_e"):
            getattr(lay, nm).weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 48).astype(np.float32)))
            getattr(lay, nm).bias.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, 48).astype(np.float32)))
    h0 = torch.from_numpy(rng.standard_normal((n, 48)).astype(np.float32))
    e0 = torch.from_numpy(rng.standard_normal((src.size, 48)).astype(np.float32))
    hd, ed = h0.to(dev).requires_grad_(True), e0.to(dev).requires_grad_(True)
    h1, e1 = lay(g, hd, ed)
    lsd = {"gnn.convs.0." + k: v.detach().cpu().double().requires_grad_(True) for k, v in lay.state_dict().items()}
    h64, e64 = h0.double().requires_grad_(True), e0.double().requires_grad_(True)
    rh, re = orc.layer_forward(lsd, 0, ts.long(), td.long(), n, h64, e64, bn)
    assert h1.shape == (n, 48) and e1.shape == (src.size, 48)
    assert_parity(h1.detach().cpu().numpy(), rh.detach().numpy(), "H=48 layer h", l2=2e-5)
    assert_parity(e1.detach().cpu().numpy(), re.detach().numpy(), "H=48 layer e", l2=2e-5)
    # ... and its backward (input and parameter gradients) under a fixed cotangent
    ch = torch.from_numpy(rng.standard_normal((n, 48)).astype(np.float32))
    ce = torch.from_numpy(rng.standard_normal((src.size, 48)).astype(np.float32))
    ((h1 * ch.to(dev)).sum() + (e1 * ce.to(dev)).sum()).backward()
    ((rh * ch.double()).sum() + (re * ce.double()).sum()).backward()
    rows = []
    _cmp("H=48 layer gh", hd.grad, h64.grad, rows)
    _cmp("H=48 layer ge", ed.grad, e64.grad, rows)
    for k, prm in lay.named_parameters():
        _cmp("H=48 layer g " + k, prm.grad, lsd["gnn.convs.0." + k].grad, rows)
    _report(rows)
    gmx = max(r_[3] for r_ in rows)
    assert all(r_[1] <= 1e-3 or r_[2] <= 2e-6 * max(gmx, 1.0) for r_ in rows), [r_ for r_ in rows if r_[1] > 1e-3]


@pytest.mark.mode_independent
def test_model_accepts_a_dgl_graph_object():
    """train.py:252 calls model(g, x, e, pe) with the DGLGraph itself: any object with edges() / num_nodes() is wrapped once
    (graph.as_assembly_graph) -- here the DGL stand-in the golden vectors were generated with; logits bit-equal to the
    AssemblyGraph built directly, the wrapper (and its index) is reused by the second call."""
    import sys
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dgl_standin"))
    try:
        import dgl
    finally:
        sys.path.pop(0)
    dev = _dev()
    src, dst, n = synth.make_graph(300, 2, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, 2)
    model = G.GraphGatedGCNModel(1, 2, 64, 16, 2, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(64, 2, 2).items()})
    model.to(dev).eval()
    e, pe = torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev)
    dg = dgl.DGLGraph(src, dst, n)
    with torch.no_grad():
        s_ref = model(G.AssemblyGraph(src, dst, n).to(dev), None, e, pe)
        s_dgl = model(dg, None, e, pe)
        w = dg._gnm_graph
        s_dgl2 = model(dg, None, e, pe)
    assert torch.equal(s_ref, s_dgl) and torch.equal(s_dgl, s_dgl2) and dg._gnm_graph is w
    assert w.device == dev and dev in w._dev_index


# -----------------------------------------------------------------------------------------
# size-independent properties at (near) BASELINE size
# -----------------------------------------------------------------------------------------

def _model_and_inputs(reads, H, L, seed, dev, permute=False):
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    src, dst, n = synth.make_graph(reads, seed, permute_edge_ids=permute)
    inp = synth.make_inputs(src, dst, n, seed)
    sd = synth.synth_state_dict(H, L, seed)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    return model, src, dst, n, inp


def test_edge_id_permutation_equivariance_and_determinism():
    """Scores are a function of the edge, not of its id: permuting edge ids permutes the scores
    (and leaves the loss and the parameter gradients unchanged up to summation order).  Running
    the same input twice is bit-identical (no atomics anywhere)."""
    import gnnome_assembly_amd as G
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(60000, 128, 4, 7, dev)
    E = src.size
    rng = np.random.default_rng(99)
    p = rng.permutation(E)
    g1 = G.AssemblyGraph(src, dst, n).to(dev)
    g2 = G.AssemblyGraph(src[p], dst[p], n).to(dev)
    pe = torch.from_numpy(inp["pe"]).to(dev)
    e1 = torch.from_numpy(inp["e"]).to(dev)
    e2 = torch.from_numpy(inp["e"][p]).to(dev)
    y1 = torch.from_numpy(inp["y"]).to(dev)
    y2 = torch.from_numpy(inp["y"][p]).to(dev)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))

    def run(g, e, y):
        model.zero_grad(set_to_none=True)
        s = model(g, None, e, pe)
        loss = crit(s.squeeze(-1), y)
        loss.backward()
        return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}

    sa, la, ga = run(g1, e1, y1)
    sb, lb, gb = run(g1, e1, y1)
    assert torch.equal(sa, sb) and la == lb and all(torch.equal(ga[k], gb[k]) for k in ga), "not deterministic"
    sc, lc, gc = run(g2, e2, y2)
    pt = torch.from_numpy(p).to(dev)
    assert_parity(sc.cpu().numpy(), sa[pt].cpu().numpy(), "permuted logits", rtol=1e-4, atol=1e-5, l2=2e-5)
    assert abs(la - lc) < 1e-6
    for k in ga:
        r = rel_l2(gc[k].cpu().numpy(), ga[k].cpu().numpy())
        assert r < 1e-3 or float((gc[k] - ga[k]).abs().max()) < GRAD_ABS_FLOOR, (k, r)


def test_chr19_scale_step_is_finite_and_self_consistent():
    """BASELINE config 2 size (R=750k: N=1.5M, E~7.5M, H=128, L=8): one fwd+bwd fits in HBM,
    everything is finite, BatchNorm invariants hold on the outputs, and the loss moves down
    along the negative gradient (a directional-derivative check of the whole backward)."""
    import gnnome_assembly_amd as G
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(750000, 128, 8, 0, dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    pe = torch.from_numpy(inp["pe"]).to(dev)
    e = torch.from_numpy(inp["e"]).to(dev)
    y = torch.from_numpy(inp["y"]).to(dev)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    s = model(g, None, e, pe)
    loss = crit(s.squeeze(-1), y)
    loss.backward()
    torch.cuda.synchronize()
    assert s.shape == (src.size, 1) and bool(torch.isfinite(s).all())
    gn2 = 0.0
    for k, prm in model.named_parameters():
        assert bool(torch.isfinite(prm.grad).all()), k
        gn2 += float((prm.grad.double() ** 2).sum())
    assert gn2 > 0
    print(f"chr19-scale: E={src.size} loss={loss.item():.6f} |g|^2={gn2:.4e} "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    # directional derivative along the gradient, central difference (the curvature term cancels, so the step can be
    # small enough for a 5 % bar and large enough for fp32 losses): L(p - eps*g) - L(p + eps*g) ~= -2*eps*|g|^2
    eps = 3e-3 / max(gn2 ** 0.5, 1e-12)
    with torch.no_grad():
        for prm in model.parameters():
            prm -= eps * prm.grad
        lm = crit(model(g, None, e, pe).squeeze(-1), y).item()
        for prm in model.parameters():
            prm += 2 * eps * prm.grad
        lp = crit(model(g, None, e, pe).squeeze(-1), y).item()
    pred = -2 * eps * gn2
    print(f"directional: L(p - eps g) - L(p + eps g) = {lm - lp:.4e} predicted = {pred:.4e}")
    assert lm < loss.item() < lp and abs((lm - lp) - pred) <= 0.05 * abs(pred) + 2e-6


@pytest.mark.default_mode_only
@pytest.mark.parametrize("case", ["r750k", "h256l16_r110k"])
def test_full_size_logits_match_the_oracle(case):
    """Parity AT THE METRIC'S SIZE (BASELINE config 2: R = 750 k, N = 1.5 M, E = 7.54 M, H = 128, L = 8): the logits of the
    HIP forward against oracle.model_forward in fp64, bar = assert_parity (rtol 1e-4, atol 1e-5, rel-L2 <= 1e-4).  The 1 k-read
    fixtures cannot exercise BatchNorm sums over 7.5 M rows, the int32 / int64 offset arithmetic or the sweep plans at one
    workgroup per CU; this does.  Run twice on the device: node ids as the generator gives them, and shuffled (the internal
    renumbering: the same oracle logits apply, logits belong to edges).
    The oracle's fp64 forward takes ~5 min of a 128-thread host, so by default the comparison is with its logits at every 97th
    edge, stored in tests/golden/fullsize_logits_r750k.npz (tests/golden/make_golden_fullsize.py; 77,735 of the 7,540,278, plus the
    norm of all of them); GNM_FULL_ORACLE=1 runs the oracle live and compares ALL E logits (profiles/r04_gputest.log has such a
    run: rel-L2 5.7e-7)."""
    import time
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    dev = _dev()
    # h256l16_r110k (round 6): the reference's DEFAULT model shape (hyperparameters.py:8,13: dim_latent 256, num_gnn_layers 16) at the true
    # chr19 size of SURVEY 8(d) -- the 256-wide fused kernels and sweeps against the fp64 oracle at size, sixteen layers deep
    (R, H, L, fixture), seed = {"r750k": (750000, 128, 8, "fullsize_logits_r750k.npz"),
                                "h256l16_r110k": (110000, 256, 16, "fullsize_logits_h256l16_r110k.npz")}[case], 0
    if not os.path.exists(os.path.join(GOLDEN, fixture)) and os.environ.get("GNM_FULL_ORACLE") != "1":
        pytest.skip(f"{fixture} not generated yet (tests/golden/make_golden_fullsize.py --case {case} on a large-memory host)")
    model, src, dst, n, inp = _model_and_inputs(R, H, L, seed, dev)
    E = int(src.size)
    model.eval()

    def hip(src_, dst_, pe_np):
        g = G.AssemblyGraph(src_, dst_, n).to(dev)
        with torch.no_grad():
            s = model(g, None, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(pe_np).to(dev))
        torch.cuda.synchronize()
        out = s.squeeze(-1).cpu().numpy()
        del g, s
        torch.cuda.empty_cache()
        return out
    s_sorted = hip(src, dst, inp["pe"])
    p = np.random.default_rng(17).permutation(n).astype(np.int32)
    pe_s = np.empty_like(inp["pe"])
    pe_s[p] = inp["pe"]
    s_shuf = hip(p[src], p[dst], pe_s)
    assert np.isfinite(s_sorted).all() and np.isfinite(s_shuf).all()
    if os.environ.get("GNM_FULL_ORACLE") == "1":
        from oracle import gatedgcn_oracle as orc
        sd = synth.synth_state_dict(H, L, seed)
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = orc.model_forward(sd_to_torch(sd, torch.float64), torch.from_numpy(src), torch.from_numpy(dst), n,
                                    torch.from_numpy(inp["e"]).double(), torch.from_numpy(inp["pe"]).double()).squeeze(-1).numpy()
        what = f"all {E} logits, live fp64 oracle ({time.perf_counter() - t0:.0f} s on {torch.get_num_threads()} threads)"
        idx = np.arange(E)
    else:
        z = np.load(os.path.join(GOLDEN, fixture))
        assert (int(z["reads"]), int(z["H"]), int(z["L"]), int(z["seed"]), int(z["edges"])) == (R, H, L, seed, E)
        idx = np.arange(0, E, int(z["stride"]))
        ref = z["logits"]
        assert ref.dtype == np.float64 and ref.size == idx.size
        # the stored norm of ALL the oracle's logits: the edges between the samples are covered in aggregate
        for s in (s_sorted, s_shuf):
            assert abs(np.linalg.norm(s.astype(np.float64)) / float(z["norm2"]) - 1.0) < 1e-5
        what = f"every {int(z['stride'])}th logit ({idx.size}), fp64 oracle fixture"
    a, b = s_sorted[idx], s_shuf[idx]
    r1, r2 = rel_l2(a, ref), rel_l2(b, ref)
    line = (f"full size E={E} N={n} H={H} L={L}: logits rel_l2 vs the oracle = {r1:.3e} (generator ids), {r2:.3e} (shuffled ids, "
            f"renumbered); max_abs {np.abs(a - ref).max():.3e} / {np.abs(b - ref).max():.3e}; {what}")
    print(line)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "logit_parity_fullsize.txt" if case == "r750k" else f"logit_parity_fullsize_{case}.txt"), "w") as f:
        f.write(line + "\n")
    assert_parity(a, ref, "full-size logits, generator ids")
    assert_parity(b, ref, "full-size logits, shuffled ids")


def test_no_grad_forward_keeps_no_activations():
    """inference.py:444-453 runs the model under torch.no_grad(): nothing may be saved for backward
    (peak memory of a forward-only pass must stay far below that of a training step)."""
    import gnnome_assembly_amd as G
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(100000, 128, 8, 1, dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    pe = torch.from_numpy(inp["pe"]).to(dev)
    e = torch.from_numpy(inp["e"]).to(dev)
    g.index()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        s1 = model(g, None, e, pe)
    torch.cuda.synchronize()
    peak_inf = torch.cuda.max_memory_allocated() - base
    torch.cuda.reset_peak_memory_stats()
    s2 = model(g, None, e, pe)
    torch.cuda.synchronize()
    peak_train = torch.cuda.max_memory_allocated() - base
    assert torch.equal(s1, s2.detach())
    eh = src.size * 128 * 4
    print(f"peak forward memory: no_grad {peak_inf / eh:.1f} [E,H] units, grad {peak_train / eh:.1f}")
    assert peak_inf < 8 * eh and peak_train > 2 * peak_inf


@pytest.mark.mode_independent
def test_feature_preparation_matches_reference(golden_dir):
    """utils.add_positional_encoding output of the reference (golden pe_pagerank.npz) and
    utils.preprocess_graph's z-score, computed on the GPU from the graph index."""
    import gnnome_assembly_amd as G
    dev = _dev()
    z = np.load(os.path.join(golden_dir, "pe_pagerank.npz"))
    g = G.AssemblyGraph(z["src"], z["dst"], int(z["n"])).to(dev)
    pe18 = G.features.positional_encoding(g, 16).cpu().numpy()
    assert np.array_equal(pe18[:, 0], z["in_deg"]) and np.array_equal(pe18[:, 1], z["out_deg"])
    assert np.abs(pe18[:, 2:] - z["pe"]).max() <= 2e-7 * np.abs(z["pe"]).max()
    rng = np.random.default_rng(3)
    ln = rng.integers(500, 30000, size=z["src"].size).astype(np.float32)
    sim = rng.random(z["src"].size).astype(np.float32)
    e = G.features.edge_features(torch.from_numpy(ln).to(dev), torch.from_numpy(sim).to(dev)).cpu()
    tl, ts = torch.from_numpy(ln), torch.from_numpy(sim)
    ref = torch.stack(((tl - tl.mean()) / tl.std(), (ts - ts.mean()) / ts.std()), 1)     # utils.py:72-74
    assert float((e - ref).abs().max()) < 5e-6


def test_training_harness_counterpart(tmp_path):
    """train.train full-graph branch counterpart: loss goes down, LR plateau scheduler and the
    checkpoint / best-model files follow the reference's schema (train.py:28-58,525-529), and a
    checkpointed state_dict loads back into the model."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth, train as T
    dev = _dev()
    samples = []
    for seed in range(3):
        src, dst, n = synth.make_graph(400, seed)
        inp = synth.make_inputs(src, dst, n, seed)
        g = G.AssemblyGraph(src, dst, n).to(dev)
        pe18 = G.features.positional_encoding(g, 16)
        assert np.abs(pe18.cpu().numpy() - inp["pe"]).max() < 1e-6
        samples.append(T.GraphSample(g, torch.from_numpy(inp["e"]).to(dev), pe18, torch.from_numpy(inp["y"]).to(dev)))
    hp = dict(num_epochs=6, dim_latent=32, num_gnn_layers=2, lr=1e-2, patience=0, decay=0.5)
    model, best, hist = T.train(samples[:2], samples[2:], out="t", hyperparameters=hp, workdir=str(tmp_path), verbose=False)
    assert hist.loss_train[-1] < hist.loss_train[0]
    assert len(hist.loss_valid) == 6 and all(np.isfinite(hist.loss_valid))
    ck = torch.load(tmp_path / "checkpoints" / "t.pt", weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optim_state_dict", "loss_train", "loss_valid"} and ck["epoch"] == 5
    m2 = G.GraphGatedGCNModel(1, 2, 32, 16, 2, 64, True, 16)
    m2.load_state_dict(ck["model_state_dict"], strict=True)
    assert list(best.keys()) == list(m2.state_dict().keys())
    acc, precision, recall, f1 = hist.metrics_train[-1]
    assert 0.0 <= acc <= 1.0 and 0.0 <= f1 <= 1.0


def test_inference_and_decode_counterpart():
    """inference.py:444-490: no_grad logits by edge id -> greedy decode; the walks equal the oracle's on the
    same logits and seed (the decode consumes scores BY EDGE ID, so this also checks the output order)."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import decode, synth
    from oracle import decode_oracle as dorc
    dev = _dev()
    src, dst, n = synth.make_graph(1500, 5, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, 5)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    model = G.GraphGatedGCNModel(1, 2, 128, 16, 2, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(128, 2, 5).items()})
    model.to(dev)
    rng = np.random.default_rng(5)
    pl, rl = rng.integers(500, 12000, src.size), rng.integers(8000, 25000, n)
    torch.manual_seed(7)
    scores, walks = decode.infer_contigs(model, g, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev),
                                         pl, rl, nb_paths=10, len_threshold=10)
    assert scores.shape == (src.size,) and len(walks) > 0
    torch.manual_seed(7)
    want = dorc.get_contigs(src, dst, n, scores.cpu().numpy(), pl, rl, nb_paths=10, len_threshold=10)
    assert walks == want
    for w in walks:                        # every step of a walk is an edge of the graph
        assert all((a, b) in set(zip(src.tolist(), dst.tolist())) for a, b in zip(w[:2], w[1:3]))


def test_minibatch_mode_counterpart(tmp_path):
    """ClusterGCN branch (train.py:282-343,428-486): a mini-batch is an induced sub-graph that goes through
    the same model; its logits equal the oracle's on that sub-graph, and the loop trains."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import cluster, synth, train as T
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    samples = []
    for seed in range(2):
        src, dst, n = synth.make_graph(3000, seed, permute_edge_ids=True)
        inp = synth.make_inputs(src, dst, n, seed)
        g = G.AssemblyGraph(src, dst, n).to(dev)
        samples.append(T.GraphSample(g, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev),
                                     torch.from_numpy(inp["y"]).to(dev)))
    # (i) one mini-batch against the oracle
    s = samples[0]
    loader = T._cluster_batches(s, 40, 7, "rcm")
    sub = next(iter(loader))
    assert 0 < sub.num_edges() < s.graph.num_edges() and sub.num_nodes() < s.graph.num_nodes()
    H, L = 128, 2
    sd = synth.synth_state_dict(H, L, 3)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    with torch.no_grad():
        got = model(sub, None, sub.edata["e"], sub.ndata["pe"])
    ssrc, sdst = sub.edges()
    with torch.no_grad():
        want = orc.model_forward(sd_to_torch(sd, torch.float64), ssrc.cpu().long(), sdst.cpu().long(), sub.num_nodes(),
                                 sub.edata["e"].cpu().double(), sub.ndata["pe"].cpu().double(), True)
    assert_parity(got.cpu().numpy(), want.numpy(), "mini-batch logits vs oracle on the induced sub-graph")
    # (ii) the loop
    hp = dict(num_epochs=3, dim_latent=32, num_gnn_layers=2, lr=1e-2, batch_size_train=4, batch_size_eval=6,
              num_parts_metis_train=112, num_parts_metis_eval=20)
    model, best, hist = T.train(samples[:1], samples[1:], out="mb", hyperparameters=hp, workdir=str(tmp_path), verbose=False)
    assert len(hist.loss_train) == 3 and all(np.isfinite(hist.loss_train)) and all(np.isfinite(hist.loss_valid))
    assert hist.loss_train[-1] < hist.loss_train[0]
    assert (tmp_path / "checkpoints" / "mb.pt").exists()


@pytest.mark.parametrize("H,L,bn", [(256, 2, True), (64, 3, False), (128, 2, False), (32, 1, True), (128, 3, True), (320, 2, True), (512, 1, True)])
def test_other_widths_and_norms_vs_oracle(H, L, bn):
    """Widths / depths / norm modes without a golden fixture: the HIP path (generic GEMM + row kernels
    for H != 128 or LayerNorm, fused kernels for H = 128 BatchNorm) against the fp64 oracle, with the
    fp32 oracle as the noise yardstick for the gradients.  (320, 512: wider than the widest kernel instantiation -- the
    layers run as 256-column problems between full-width dense products, 320 zero-padded to 512: engine.WIDE_CHUNK.)"""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    src, dst, n = synth.make_graph(700, seed=H + L, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, seed=H)
    sd = synth.synth_state_dict(H, L, seed=L)
    z = dict(src=src, dst=dst, n=n, e_raw=inp["e"], pe=inp["pe"], y=inp["y"], pos_weight=inp["pos_weight"])
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, bn, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    s = model(g, None, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev))
    loss = G.BCEWithLogitsLoss(float(inp["pos_weight"]))(s.squeeze(-1), torch.from_numpy(inp["y"]).to(dev))
    loss.backward()
    p64 = sd_to_torch(sd, torch.float64, requires_grad=True)
    s64 = orc.model_forward(p64, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).double(),
                            torch.from_numpy(inp["pe"]).double(), bn)
    l64 = orc.bce_loss(s64, torch.from_numpy(inp["y"]).double(), float(inp["pos_weight"]))
    l64.backward()
    assert_parity(s.detach().cpu().numpy(), s64.detach().numpy(), f"H={H} L={L} bn={bn} logits")
    assert abs(loss.item() - l64.item()) < 1e-5
    g32 = None if bn else _oracle_grads(z, sd, torch.float32, bn)
    bad = []
    for k, prm in model.named_parameters():
        got, want = prm.grad.detach().cpu().double().numpy(), p64[k].grad.numpy()
        r, r32 = rel_l2(got, want), (None if bn else rel_l2(g32[k], want))
        if not _grad_ok(r, float(np.abs(got - want).max()), GRAD_ABS_FLOOR, r32):
            bad.append((k, r, r32))
    if bad and bn:      # BatchNorm: only relu-kink flips may explain a miss (see test_model_matches_golden); no noise clause
        brows, bgmax = _branch_exact_rows(src, dst, n, inp["e"], inp["pe"], inp["y"], float(inp["pos_weight"]), sd, L, dev)
        _branch_exact_or_fail(bad, {r[0]: r for r in brows}, bgmax, f"H={H} L={L}")
        bad = []
    assert not bad, bad


from helpers import branch_exact_rows as _branch_exact_rows  # noqa: E402  (shared with __graft_entry__.smoke)


BRANCH_L2 = 5e-5


@pytest.mark.parametrize("case", ["small_h128l8_s0.npz", "small_h128l8_s1.npz", "small_h64l1_s1.npz", "tiny_h64l1_s0.npz",
                                  "synth_h256l2", "synth_h64l4", "synth_h128l3", "synth_h32l1"])
def test_gradients_exact_for_the_branch_taken(case):
    """Gradient parity without the relu-kink ambiguity: the fp64 oracle backward is evaluated on the
    SAME relu branches the device took (see _device_masks); every parameter gradient must then agree to
    fp32 round-off (rel-L2 <= 5e-5), including the B_1/B_2/B_3 tensors whose plain comparison is limited
    by single-element branch flips of either side."""
    from gnnome_assembly_amd import synth
    dev = _dev()
    if case.startswith("synth"):
        H, L = {"synth_h256l2": (256, 2), "synth_h64l4": (64, 4), "synth_h128l3": (128, 3), "synth_h32l1": (32, 1)}[case]
        src, dst, n = synth.make_graph(700, seed=H + L, permute_edge_ids=True)
        inp = synth.make_inputs(src, dst, n, seed=H)
        sd = synth.synth_state_dict(H, L, seed=L)
        e_raw, pe, y, pw = inp["e"], inp["pe"], inp["y"], float(inp["pos_weight"])
    else:
        z, sd, H, L, bn = load_case(case)
        src, dst, n = z["src"], z["dst"], int(z["n"])
        e_raw, pe, y, pw = z["e_raw"], z["pe"], z["y"], float(z["pos_weight"])
    rows, gmax = _branch_exact_rows(src, dst, n, e_raw, pe, y, pw, sd, L, dev)
    _report(rows, f"branch_{case}.txt")
    bad = [r for r in rows if r[1] > BRANCH_L2 and r[2] > max(GRAD_ABS_FLOOR, 1e-6 * gmax)]
    assert not bad, bad


def test_full_size_forward_is_edge_id_order_equivariant():
    """BASELINE config 2 size (E ~ 7.5 M): relabelling the edges permutes the logits and nothing else, up
    to fp32 round-off (edges of one destination keep their caller order inside the internal layout, so
    the order of a node's segmented sums follows the labels), and a repeated run is bit-identical."""
    import gnnome_assembly_amd as G
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(750000, 128, 8, 0, dev)
    e = torch.from_numpy(inp["e"]).to(dev)
    pe = torch.from_numpy(inp["pe"]).to(dev)
    with torch.no_grad():
        s0 = model(G.AssemblyGraph(src, dst, n).to(dev), None, e, pe)
        perm = torch.randperm(src.size, generator=torch.Generator().manual_seed(1))
        pn = perm.numpy()
        s1 = model(G.AssemblyGraph(src[pn], dst[pn], n).to(dev), None, e[perm.to(dev)], pe)
        s2 = model(G.AssemblyGraph(src, dst, n).to(dev), None, e, pe)
    assert bool(torch.isfinite(s0).all()) and torch.equal(s2, s0)
    want = s0[perm.to(dev)]
    r = float((s1 - want).double().norm() / want.double().norm())
    print(f"full-size permutation equivariance: rel_l2 {r:.2e}, max abs {float((s1 - want).abs().max()):.2e}")
    assert r <= 2e-5 and float((s1 - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-5


def test_train_loop_matches_the_reference_loop(tmp_path, golden_dir):
    """gnnome_assembly_amd.train.train (the build's own LOOP, not a hand-rolled one) against the reference's
    full-graph training loop run on the reference's own model (tests/golden/make_golden_train.py: train.py:181,
    195-212,237-258,346-353,385-411,525-529): same seeded initial weights, same shuffled graph order, the per-step
    loss sequence, the per-epoch train / validation losses, TP/TN/FP/FN totals, the ReduceLROnPlateau LR
    sequence (it halves twice in this run) and the best epoch."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth, train as T
    dev = _dev()
    z = np.load(os.path.join(golden_dir, "train_loop_h128l2.npz"))
    hp = {str(k): float(v) for k, v in zip(z["hp_keys"], z["hp_vals"])}
    for k in ("seed", "num_epochs", "dim_latent", "node_features", "edge_features", "hidden_edge_features", "hidden_edge_scores",
              "num_gnn_layers", "nb_pos_enc", "batch_size_train", "batch_size_eval", "patience"):
        hp[k] = int(hp[k])
    hp["batch_norm"] = bool(hp["batch_norm"])

    def sample(spec):
        reads, seed, permute = (int(v) for v in spec)
        src, dst, n = synth.make_graph(reads, seed, permute_edge_ids=bool(permute))
        inp = synth.make_inputs(src, dst, n, seed)
        g = G.AssemblyGraph(src, dst, n).to(dev)
        return T.GraphSample(g, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev),
                             torch.from_numpy(inp["y"]).to(dev))
    train = [sample(z[f"train{i}"]) for i in range(2)]
    valid = [sample(z["valid0"])]
    # the seeded initialisation of the build's model class == the reference class's (same creation order)
    torch.manual_seed(hp["seed"])
    m0 = G.GraphGatedGCNModel(hp["node_features"], hp["edge_features"], hp["dim_latent"], hp["hidden_edge_features"],
                              hp["num_gnn_layers"], hp["hidden_edge_scores"], hp["batch_norm"], hp["nb_pos_enc"])
    for k, v in m0.state_dict().items():
        assert np.array_equal(v.numpy().reshape(-1)[::13], z["init/" + k]), f"seeded init differs: {k}"
    model, best, hist = T.train(train, valid, out="pin", hyperparameters=hp, workdir=str(tmp_path), verbose=False)
    assert abs(T.pos_to_neg_ratio(train) - float(z["ratio64"])) < 1e-6                      # train.py:181
    assert hist.step_graph == z["step_graph64"].tolist()                                    # random.shuffle order
    got, w64, w32 = np.array(hist.step_losses), z["step_losses64"], z["step_losses32"]
    noise = np.abs(w32 - w64) / w64                  # how far the reference's own fp32 run drifts from its fp64 run
    rel = np.abs(got - w64) / w64
    print("per-step loss rel. error vs reference fp64:", np.array2string(rel, precision=2),
          "reference fp32:", np.array2string(noise, precision=2))
    assert np.all(rel <= np.maximum(1e-4, 5 * noise)), (got, w64)
    for name, g_, ref in (("train", hist.loss_train, z["loss_train64"]), ("valid", hist.loss_valid, z["loss_valid64"])):
        assert np.allclose(g_, ref, rtol=1e-4, atol=0), (name, g_, ref)
    assert hist.lr == z["lr64"].tolist() and hist.final_lr == float(z["final_lr64"])       # ReduceLROnPlateau
    assert hist.best_epoch == int(z["best_epoch64"])
    # TP/TN/FP/FN: threshold statistics of the logits after k Adam steps at lr = 2e-2.  A prediction within the
    # accumulated round-off of 0.5 falls either way, and that band widens with every step (Adam normalises each gradient
    # entry, so fp32-level differences between two correct evaluations are re-amplified each step: the per-step losses
    # above stay within 1e-4, the last epoch's logits differ by ~1e-3).  The first epoch must match the reference's
    # fp64 run to the band its own fp32 run shows (+2 edges); later epochs additionally get 0.1 % of the edges counted
    # (which of two equally valid fp32 backward kernels is used moves epoch 4 by 3-8 edges of 10,794).
    for name, g_, r64, r32 in (("train", hist.tfpn_train, z["tfpn_train64"], z["tfpn_train32"]),
                               ("valid", hist.tfpn_valid, z["tfpn_valid64"], z["tfpn_valid32"])):
        g_ = np.array(g_)
        assert g_.shape == r64.shape and np.array_equal(g_.sum(1), r64.sum(1))
        slack = np.full(g_.shape, 2.0)
        slack[1:] = np.maximum(2.0, np.ceil(1e-3 * r64.sum(1, keepdims=True)[1:]))
        assert np.all(np.abs(g_ - r64) <= 3 * np.abs(r32 - r64) + slack), (name, g_.tolist(), r64.tolist())
    # final weights after 8 Adam steps (strided sample) vs the fp64 run.  Adam normalises every gradient entry by
    # its own running magnitude, so an entry that is round-off-sized in one step moves by up to lr in either
    # direction: single elements differ by O(lr) between ANY two fp32 evaluations; the tensors as a whole agree
    got_all = np.concatenate([v.detach().cpu().double().numpy().reshape(-1)[::13] for v in model.state_dict().values()])
    want_all = np.concatenate([z["final/" + k] for k in model.state_dict()])
    r = rel_l2(got_all, want_all)
    print(f"final weights vs reference fp64: rel_l2 {r:.2e}, max abs {np.abs(got_all - want_all).max():.2e}")
    assert r <= 2e-2 and np.abs(got_all - want_all).max() <= 4 * hp["lr"]


def test_flat_gradient_fast_path_equals_autograd_accumulation():
    """With parameters flattened (models.flatten_parameters) and a zeroed dp.FlatGradients the backward kernels write
    every gradient straight into the flat buffer (no autograd accumulation kernels).  Same kernels, so the
    result is bit-identical to the ordinary .grad accumulation; without zero_() in between a second backward
    accumulates as autograd does."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(3000, 128, 3, 4, dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    crit(model(g, None, e, pe).squeeze(-1), y).backward()                 # ordinary autograd accumulation
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.flatten_parameters()
    flat = dp.FlatGradients(model.parameters(), direct_write=True)          # the fast path is an opt-in
    named = dict(model.named_parameters())
    a = [named[f"gnn.convs.1.{k}.weight"].grad for k in ("A_1", "A_2", "A_3", "B_1", "B_2")]
    assert all(x.data_ptr() + x.numel() * 4 == y_.data_ptr() for x, y_ in zip(a[:-1], a[1:]))   # one [5H,H] block
    flat.zero_()
    assert flat.fresh
    crit(model(g, None, e, pe).squeeze(-1), y).backward()
    assert not flat.fresh
    assert all(torch.equal(p.grad, ref[k]) for k, p in named.items())
    crit(model(g, None, e, pe).squeeze(-1), y).backward()                 # not zeroed: accumulates (autograd route)
    assert all(torch.equal(p.grad, 2 * ref[k]) for k, p in named.items())
    assert all(flat.flat.data_ptr() <= p.grad.data_ptr() < flat.flat.data_ptr() + flat.flat.numel() * 4 for p in named.values())
    # a gradient written by anything else than the model's backward leaves the buffer "not fresh": the next backward
    # accumulates instead of overwriting (all_reduce_mean on a padding step did not clear the flag)
    flat.zero_()
    flat.all_reduce_mean(contributed=False)
    assert not flat.fresh
    # a parameter hook switches the fast path off (hooks only see gradients that travel through autograd) ...
    flat.zero_()
    seen = []
    hnd = named["predictor.W2.bias"].register_hook(lambda gr: seen.append(float(gr.sum())))
    crit(model(g, None, e, pe).squeeze(-1), y).backward()
    hnd.remove()
    assert seen and not flat.fresh and all(torch.equal(p.grad, ref[k]) for k, p in named.items())
    # ... and without the opt-in every gradient takes the autograd route: autograd.grad returns them
    plain = dp.FlatGradients(model.parameters())
    plain.zero_()
    gs = torch.autograd.grad(crit(model(g, None, e, pe).squeeze(-1), y), list(named.values()))
    assert all(torch.equal(a_, ref[k]) for a_, k in zip(gs, named))


def test_lean_activation_mode_is_bit_identical_and_smaller():
    """engine.set_activation_mode("lean"): P [N,5H] and t [E,H] are not kept for the backward but rebuilt by the
    kernels that made them.  Same kernels on the same inputs: logits, loss and every gradient bit-identical to
    the default mode; the peak memory of a training step drops by about 2 of the ~5.3 [E,H]-sized tensors a
    layer keeps."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(60000, 128, 6, 2, dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    g.index()

    def run(mode):
        engine.set_activation_mode(mode)
        try:
            model.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            s = model(g, None, e, pe)
            loss = crit(s.squeeze(-1), y)
            loss.backward()
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated() - base
            return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}, peak
        finally:
            engine.set_activation_mode("saved")
    s0, l0, g0, p0 = run("saved")
    s1, l1, g1, p1 = run("lean")
    assert torch.equal(s0, s1) and l0 == l1 and all(torch.equal(g0[k], g1[k]) for k in g0)
    unit = 4.0 * src.size * 128
    print(f"peak memory of one training step: saved {p0 / unit:.1f} [E,H] units, lean {p1 / unit:.1f}")
    assert p1 < p0 - 6 * 1.0 * unit          # ~2 units per layer (t, and P = 5N/E units), 6 layers; the backward rebuilds
                                             # them for at most two layers at a time


def test_chained_backward_matches_the_layer_by_layer_backward():
    """engine.CHAIN (default, bf16x3 mode): layer i's fused edge backward and layer i-1's by-destination pass run in
    one kernel.  Same arithmetic as L x layer_backward up to the order of the fp32 partial sums of the weight
    gradients: every gradient agrees to 2e-5 (the B_3 weight gradient: other row partition of the same products),
    the loss and the logits are bit-identical (the forward does not change)."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(30000, 128, 4, 3, dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))

    def run(chain):
        with engine.options(CHAIN=chain):
            model.zero_grad(set_to_none=True)
            s = model(g, None, e, pe)
            loss = crit(s.squeeze(-1), y)
            loss.backward()
            torch.cuda.synchronize()
            return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}
    s0, l0, g0 = run(False)
    s1, l1, g1 = run(True)
    assert torch.equal(s0, s1) and l0 == l1
    gmax = max(float(v.abs().max()) for v in g0.values())
    bad = []
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        r = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if r > 2e-5 and float((a - b).abs().max()) > 1e-6 * gmax:
            bad.append((k, r))
    assert not bad, bad


# -----------------------------------------------------------------------------------------
# round 5: the node side (fused conversion / BatchNorm_h sums)
# -----------------------------------------------------------------------------------------

@pytest.mark.default_mode_only
def test_fused_node_backward_kernels_match_the_separate_launches():
    """gnm_tn128_bgrad = gnm_node_bgrad + gnm_tn128 over gB1h | gB2h (the groups it forms are written to gP for the
    projection backward); gnm_node_proj_bwd_nn_stats = gnm_node_proj_bwd_nn + gnm_node_bwd_stats of the layer below.
    Same expressions: gP columns and gh_in bit for bit, weight gradients / BatchNorm sums to summation order."""
    import ctypes as C
    from gnnome_assembly_amd import engine, _lib
    dev = _dev()
    lib = _lib.load()
    H = 128
    rng = np.random.default_rng(22)
    f = lambda *sh, s=1.0: torch.from_numpy((rng.standard_normal(sh) * s).astype(np.float32)).to(dev)      # noqa: E731
    for N, pitch2 in ((50021, True), (4096, False)):
        UT, DT = f(N, 2 * H), f(N, 2 * H)
        Ud, Td = (DT[:, :H], DT[:, H:]) if pitch2 else (DT[:, :H].contiguous(), DT[:, H:].contiguous())
        stat_e, bstat_e, gamma_e = f(4, H).abs() + 0.1, f(2, H, s=0.1), f(H).abs() + 0.5
        deg = rng.integers(0, 9, size=(2, N))
        in_ptr = torch.from_numpy(np.concatenate(([0], np.cumsum(deg[0]))).astype(np.int32)).to(dev)
        out_ptr = torch.from_numpy(np.concatenate(([0], np.cumsum(deg[1]))).astype(np.int32)).to(dev)
        h, W5, gh_out, z = f(N, H), f(5 * H, H, s=0.09), f(N, H, s=1e-3), f(N, H)
        stat_h = f(4, H)
        gP0 = f(N, 5 * H, s=1e-3)
        gP1 = gP0.clone()
        gP1[:, 3 * H:] = float("nan")          # the fused kernel must write every element of the two groups
        st = engine._stream()
        sc = engine.scratch(dev)
        needt, needp = lib.gnm_tn128_workspace_bytes(), lib.gnm_node_proj_bwd_workspace_bytes(5 * H)
        ws = sc.ws(max(needt, needp))
        # separate launches
        engine._call("gnm_node_bgrad", N, H, engine._ptr(stat_e), engine._ptr(bstat_e), engine._ptr(gamma_e), engine._ptr(in_ptr),
                     engine._ptr(out_ptr), engine._ptr(UT), engine._ptr(Ud), engine._ptr(Td), Ud.stride(0), engine._ptr(gP0), st)
        gW0, gb0 = torch.empty(2 * H, H, device=dev), torch.empty(2 * H, device=dev)
        engine.tn128(N, gP0[:, 3 * H:], 5 * H, 2, h, gW0, gb0, sc.partials, ws, needt)
        for _rep in range(2):
            gW1, gb1 = torch.empty(2 * H, H, device=dev), torch.empty(2 * H, device=dev)
            gP1[:, 3 * H:] = float("nan")
            engine._call("gnm_tn128_bgrad", N, H, engine._ptr(UT), engine._ptr(Ud), engine._ptr(Td), Ud.stride(0),
                         engine._ptr(stat_e), engine._ptr(bstat_e), engine._ptr(gamma_e), engine._ptr(in_ptr), engine._ptr(out_ptr),
                         engine._ptr(gP1), engine._ptr(h),
                         engine._ptr(gW1), engine._ptr(gb1), engine._ptr(sc.partials), engine._ptr(ws), needt, st)
            assert bool(torch.isfinite(gP1).all())
            d = (gP1[:, 3 * H:] - gP0[:, 3 * H:]).abs().max().item()
            assert d <= 1e-6 * gP0[:, 3 * H:].abs().max().item(), f"formed gB1h | gB2h differ from gnm_node_bgrad: {d}"
            assert float((gW1.double() - gW0.double()).norm() / gW0.double().norm()) < 1e-6
            assert float((gb1.double() - gb0.double()).norm() / gb0.double().norm()) < 1e-6
        # projection backward with the BatchNorm_h sums of the layer below
        gh0, gh1 = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
        engine._call("gnm_node_proj_bwd_nn", N, H, 5 * H, engine._ptr(gP0), engine._ptr(W5), engine._ptr(gh_out), engine._ptr(gh0),
                     engine._ptr(ws), needp, st)
        nb0 = C.c_int(0)
        engine._call("gnm_node_bwd_stats", N, H, engine._ptr(z), engine._ptr(stat_h), engine._ptr(gh0), engine._ptr(sc.partials),
                     C.byref(nb0), st)
        b0, gg0, gbt0 = engine.bn_bwd_finalize(sc.partials, nb0.value, N, H, dev)
        nb1 = C.c_int(0)
        engine._call("gnm_node_proj_bwd_nn_stats", N, H, 5 * H, engine._ptr(gP0), engine._ptr(W5), engine._ptr(gh_out),
                     engine._ptr(gh1), engine._ptr(z), engine._ptr(stat_h), engine._ptr(sc.partials), C.byref(nb1),
                     engine._ptr(ws), needp, st)
        b1, gg1, gbt1 = engine.bn_bwd_finalize(sc.partials, nb1.value, N, H, dev)
        assert torch.equal(gh0, gh1)
        for a_, b_ in ((b0, b1), (gg0, gg1), (gbt0, gbt1)):
            assert float((a_.double() - b_.double()).abs().max()) <= 1e-6 * float(a_.double().abs().max()) + 1e-12
        torch.cuda.synchronize()


@pytest.mark.default_mode_only
@pytest.mark.parametrize("ids", ["sorted", "shuffled"])
def test_round5_node_side_schedule_matches_the_round4_schedule(ids):
    """engine.NODE_FUSED (default) against the round-4 schedule (gnm_node_bgrad and gnm_node_bwd_stats as launches of their own): the
    forward is the same code, every gradient agrees to summation order, two runs are bit-identical; under lean activations (no side
    stream: the same launches back to back) and with the deferred weight-gradient kernel issued at the other point, bit for bit."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(40000, 128, 4, 9, dev)
    pe_np, e_np = inp["pe"], inp["e"]
    if ids == "shuffled":
        p = np.random.default_rng(6).permutation(n).astype(np.int32)
        src, dst = p[src], p[dst]
        pe_s = np.empty_like(pe_np)
        pe_s[p] = pe_np
        pe_np = pe_s
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = torch.from_numpy(e_np).to(dev), torch.from_numpy(pe_np).to(dev), torch.from_numpy(inp["y"]).to(dev)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))

    def run(**opts):
        with engine.options(**opts):
            model.zero_grad(set_to_none=True)
            s = model(g, None, e, pe)
            loss = crit(s.squeeze(-1), y)
            loss.backward()
            torch.cuda.synchronize()
            return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}
    s0, l0, g0 = run(NODE_FUSED=False)
    s1, l1, g1 = run()
    s2, l2, g2 = run()
    assert torch.equal(s0, s1) and l0 == l1, "the backward schedule must not change the forward"
    assert all(torch.equal(g1[k], g2[k]) for k in g1), "not run-to-run deterministic"
    gmax = max(float(v.abs().max()) for v in g0.values())
    bad = []
    for k in g1:
        a, b = g1[k].double(), g0[k].double()
        r = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if r > 2e-6 and float((a - b).abs().max()) > 1e-7 * gmax:
            bad.append((k, r))
    assert not bad, ("round 5 vs round 4", bad)
    s6, _, g6 = run(ACTIVATIONS="lean")
    assert torch.equal(s6, s1) and all(torch.equal(g6[k], g1[k]) for k in g1), "lean activations must be bit-identical"
    for at in ("next", "now"):        # the default is "auto" = by graph size (engine.tn_at): both placements explicitly
        _, _, g7 = run(TN_AT=at)
        assert all(torch.equal(g7[k], g1[k]) for k in g1), "where the deferred weight-gradient kernel runs must not matter"


@pytest.mark.default_mode_only
def test_full_size_gradients_match_the_oracle():
    """Backward parity AT THE METRIC'S SIZE (BASELINE config 2: R = 750 k, N = 1.5 M, E = 7.54 M, H = 128, L = 8): the loss and
    EVERY parameter gradient (826,033 values in 138 tensors) of one HIP training step against the fp64 autograd oracle
    (oracle.bce_loss(oracle.model_forward(...)).backward() = the reference's loss.backward(), train.py:253-257), stored in
    tests/golden/fullsize_grads_r750k.npz by tests/golden/make_golden_fullsize.py --grads (22 min on a 128-thread host).  Bar: the
    same _grad_ok clauses as the small fixtures, minus the fp32-noise clause (no fp32 oracle run at this size): rel-L2 <= 2e-4 per
    tensor or under the absolute floor; the tally is printed and written to gpurun_out/grad_parity_fullsize.txt.  Exercises what
    the 1 k-read fixtures cannot: the two-sided backward sweep at one workgroup per CU, the plan's slot discipline over 7.5 M
    rows, int64 offsets, the BatchNorm-backward sums over E and N.  Run twice: generator node ids, and shuffled ids (renumbered
    by the index; parameter gradients do not depend on the numbering)."""
    import gnnome_assembly_amd as G
    dev = _dev()
    R, H, L, seed = 750000, 128, 8, 0
    zf = np.load(os.path.join(GOLDEN, "fullsize_grads_r750k.npz"))
    model, src, dst, n, inp = _model_and_inputs(R, H, L, seed, dev)
    E = int(src.size)
    assert (int(zf["reads"]), int(zf["H"]), int(zf["L"]), int(zf["seed"]), int(zf["edges"])) == (R, H, L, seed, E)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    e, y = torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["y"]).to(dev)

    def hip(src_, dst_, pe_np):
        g = G.AssemblyGraph(src_, dst_, n).to(dev)
        model.zero_grad(set_to_none=True)
        s = model(g, None, e, torch.from_numpy(pe_np).to(dev))
        loss = crit(s.squeeze(-1), y)
        loss.backward()
        torch.cuda.synchronize()
        out = loss.item(), {k: v.grad.detach().cpu().double().numpy() for k, v in model.named_parameters()}
        del g, s, loss
        torch.cuda.empty_cache()
        return out
    runs = {"generator ids": hip(src, dst, inp["pe"])}
    p = np.random.default_rng(17).permutation(n).astype(np.int32)
    pe_s = np.empty_like(inp["pe"])
    pe_s[p] = inp["pe"]
    runs["shuffled ids"] = hip(p[src], p[dst], pe_s)
    gmax = max(float(zf["norm::" + k]) for k in runs["generator ids"][1])
    lines, bad = [], []
    for what, (loss, grads) in runs.items():
        assert abs(loss - float(zf["loss"])) <= 1e-5 * abs(float(zf["loss"])), (what, loss, float(zf["loss"]))
        tally = {"l2": 0, "floor": 0, "miss": 0}
        worst = ("", 0.0)
        for k, got in grads.items():
            want = zf["grad::" + k].astype(np.float64)
            assert got.shape == want.shape, k
            r = rel_l2(got, want)
            mx = float(np.abs(got - want).max())
            clause = "l2" if r <= GRAD_L2 else "floor" if mx <= GRAD_ABS_FLOOR * max(gmax, 1.0) else "miss"
            tally[clause] += 1
            tally_clause(clause)
            if clause == "l2" and r > worst[1]:
                worst = (k, r)
            if clause == "miss":
                bad.append((what, k, r, mx))
            lines.append(f"{what:14s} {k:34s} rel_l2={r:.3e} max_abs={mx:.3e} ref_norm={float(zf['norm::' + k]):.3e} {clause}")
        head = (f"full size E={E} N={n} H={H} L={L} [{what}]: loss {loss:.9f} (oracle {float(zf['loss']):.9f}); {len(grads)} gradient "
                f"tensors: {tally['l2']} within rel-L2 {GRAD_L2:g}, {tally['floor']} under the absolute floor, {tally['miss']} missed; "
                f"worst rel-L2 among the former {worst[1]:.3e} ({worst[0]})")
        print(head)
        lines.insert(0, head)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "grad_parity_fullsize.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert not bad, bad


@pytest.mark.default_mode_only
@pytest.mark.parametrize("ids", ["sorted", "shuffled"])
def test_two_sided_sweep_matches_the_separate_by_source_pass(ids):
    """engine.TWO_SIDED (default): the chained kernel also forms layer i-1's by-SOURCE sums (gA2h, Us, Ts) through the
    graph's sweep plan, edge_bwd_src_k's three re-read [E,H] streams shrink to a gather over the nodes the plan does
    not serve + the m1 / m2 conversion.  Same per-edge terms, another (fixed) order of the additions inside a
    source's sum: every gradient agrees with the separate-pass schedule to 2e-5, two runs are bit-identical, the
    forward does not change.  Node ids as the generator gives them and shuffled (renumbered by the index)."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(40000, 128, 4, 7, dev)
    pe_np, e_np = inp["pe"], inp["e"]
    if ids == "shuffled":
        p = np.random.default_rng(3).permutation(n).astype(np.int32)
        src, dst = p[src], p[dst]
        pe_s = np.empty_like(pe_np)
        pe_s[p] = pe_np
        pe_np = pe_s
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = torch.from_numpy(e_np).to(dev), torch.from_numpy(pe_np).to(dev), torch.from_numpy(inp["y"]).to(dev)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    plan = g.sweep_plan(dev)
    assert plan is not None and 0 < plan["nfix"] < 0.2 * n, plan and plan["nfix"]
    print(f"sweep plan [{ids}]: {plan['nfix']} of {n} nodes left to the fix-up pass, peak live slots {plan['peak_live']}")

    def run(two_sided):
        with engine.options(TWO_SIDED=two_sided):
            model.zero_grad(set_to_none=True)
            s = model(g, None, e, pe)
            loss = crit(s.squeeze(-1), y)
            loss.backward()
            torch.cuda.synchronize()
            return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}
    s0, l0, g0 = run(False)
    s1, l1, g1 = run(True)
    s2, l2, g2 = run(True)
    assert torch.equal(s0, s1) and l0 == l1
    assert all(torch.equal(g1[k], g2[k]) for k in g1), "two-sided sweep is not run-to-run deterministic"
    gmax = max(float(v.abs().max()) for v in g0.values())
    bad = []
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        r = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if r > 2e-5 and float((a - b).abs().max()) > 1e-6 * gmax:
            bad.append((k, r))
    assert not bad, bad


@pytest.mark.default_mode_only
@pytest.mark.parametrize("ids", ["sorted", "shuffled"])
def test_two_sided_forward_sweep_matches_the_separate_passes(ids):
    """engine.TWO_SIDED_FWD (default): gnm_edge_gate2_fwd forms the by-destination AND the by-source gated means in one
    sweep (sweep plan over the two-workgroups-per-CU partition) instead of edge_gate_fwd + node_agg_src_fwd re-reading
    e_out.  Same per-edge terms, another fixed order inside a node's sum: logits within 2e-6 (rel-L2) of the
    separate-pass forward, gradients within 2e-5, two runs bit-identical."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(16000, 128, 4, 11, dev)
    pe_np, e_np = inp["pe"], inp["e"]
    if ids == "shuffled":
        p = np.random.default_rng(5).permutation(n).astype(np.int32)
        src, dst = p[src], p[dst]
        pe_s = np.empty_like(pe_np)
        pe_s[p] = pe_np
        pe_np = pe_s
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = torch.from_numpy(e_np).to(dev), torch.from_numpy(pe_np).to(dev), torch.from_numpy(inp["y"]).to(dev)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    plan = g.sweep_plan(dev, 2)
    assert plan is not None and 0 < plan["nfix"] < 0.25 * n
    print(f"forward sweep plan [{ids}]: {plan['nfix']} of {n} nodes left to the fix-up pass, peak live slots {plan['peak_live']}")

    def run(two_sided):
        with engine.options(TWO_SIDED_FWD=two_sided):
            model.zero_grad(set_to_none=True)
            s = model(g, None, e, pe)
            loss = crit(s.squeeze(-1), y)
            loss.backward()
            torch.cuda.synchronize()
            return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}
    s0, l0, g0 = run(False)
    s1, l1, g1 = run(True)
    s2, l2, g2 = run(True)
    assert torch.equal(s1, s2) and l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1), "not run-to-run deterministic"
    r = rel_l2(s1.cpu().numpy(), s0.cpu().numpy())
    print(f"two-sided forward vs separate passes [{ids}]: logits rel_l2 = {r:.2e}")
    assert r <= 2e-6 and abs(l1 - l0) <= 1e-6 * abs(l0)
    # the forward itself moves by ~1e-7, so the gradients are compared at the bar they have against the oracle (GRAD_L2):
    # the BatchNorm-backward differences amplify a forward perturbation the way they amplify fp32 round-off
    gmax = max(float(v.abs().max()) for v in g0.values())
    bad = []
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        rr = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if rr > GRAD_L2 and float((a - b).abs().max()) > 1e-6 * gmax:
            bad.append((k, rr))
    if bad:
        # a forward that moved by 1e-6 takes other relu branches at a few elements; a tensor that then differs by more than the
        # plain bar must be the EXACT gradient of its own branches under BOTH schedules (fp64 backward on the device's branches,
        # rel-L2 <= BRANCH_L2) -- the bar the oracle tests hold such a tensor to, not a wider one
        from gnnome_assembly_amd import synth
        sd = synth.synth_state_dict(128, 4, 11)
        for two_sided in (False, True):
            with engine.options(TWO_SIDED_FWD=two_sided):
                brows, bgmax = _branch_exact_rows(src, dst, n, e_np, pe_np, inp["y"], float(inp["pos_weight"]), sd, 4, dev)
            tally_clause("miss", len(bad))          # moved to "branch_exact" by the call below, or the test fails
            _branch_exact_or_fail(bad, {r_[0]: r_ for r_ in brows}, bgmax, f"TWO_SIDED_FWD={two_sided}", floor=0.0)


@pytest.mark.default_mode_only
@pytest.mark.parametrize("H", [128, 96])
def test_layernorm_two_sided_forward_sweep_matches_the_separate_passes(H):
    """batch_norm = False (nn.LayerNorm, gated_gcn_full.py:57-59) at H = 128 -- and at 96, which runs zero-padded to 128 with the row
    statistics over the 96 real channels: gnm_ln_edge_gate2_fwd forms e_out, the by-destination AND the by-source gated means in one
    sweep (LayerNorm has no global barrier: the row statistics are taken inside the sweep) instead of gnm_ln_edge_gate_fwd +
    gnm_node_agg_src_fwd.  Same per-edge expressions (e_out bit-identical), another fixed order inside a node's sums: logits within
    2e-6 (rel-L2) of the separate-pass forward and of the fp64 oracle's bar, gradients at the bar they have against the oracle, two
    runs bit-identical."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine, synth
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    L = 3
    src, dst, n = synth.make_graph(12000, 13)
    inp = synth.make_inputs(src, dst, n, 13)
    sd = synth.synth_state_dict(H, L, 13)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, False, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    plan = g.sweep_plan(dev, 2)
    assert plan is not None and 0 < plan["nfix"] < 0.3 * n

    def run(two_sided):
        with engine.options(TWO_SIDED_FWD=two_sided):
            model.zero_grad(set_to_none=True)
            s = model(g, None, e, pe)
            loss = crit(s.squeeze(-1), y)
            loss.backward()
            torch.cuda.synchronize()
            return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}
    engine.profile_ops(True)
    s1, l1, g1 = run(True)
    ops = engine.profile_ops(False)
    assert "gnm_ln_edge_gate2_fwd" in ops and "gnm_node_agg_src_fwd" not in ops        # the sweep really is what ran
    s0, l0, g0 = run(False)
    s2, l2, g2 = run(True)
    assert torch.equal(s1, s2) and l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1), "not run-to-run deterministic"
    r = rel_l2(s1.cpu().numpy(), s0.cpu().numpy())
    print(f"LayerNorm two-sided forward vs separate passes [H={H}]: logits rel_l2 = {r:.2e}")
    assert r <= 2e-6 and abs(l1 - l0) <= 1e-6 * abs(l0)
    p64 = sd_to_torch(sd, torch.float64)
    s64 = orc.model_forward(p64, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).double(),
                            torch.from_numpy(inp["pe"]).double(), False)
    assert_parity(s1.cpu().numpy(), s64.numpy(), f"LayerNorm two-sided forward H={H} vs the fp64 oracle")
    gmax = max(float(v.abs().max()) for v in g0.values())
    bad = []
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        rr = float((a - b).norm() / b.norm().clamp_min(1e-30))
        if rr > GRAD_L2 and float((a - b).abs().max()) > 1e-6 * gmax:
            bad.append((k, rr))
    assert not bad, bad


@pytest.mark.default_mode_only
def test_wide_layers_fused_kernels_and_sweeps_match_the_generic_route():
    """H = 256 (the reference's default dim_latent, hyperparameters.py:8).  Default: t + BatchNorm sums from the fused forward
    kernel (edge_t32_h256_k), gt and ge_in from one pass (edge_gt_nn_h256_k), the fused edge-encoder kernels, and the two-sided
    sweeps of the 128-wide path run once per 128-column half with row pitch 256.  engine.WIDE_FUSED / TWO_SIDED / TWO_SIDED_FWD
    off = the generic route (split-mode GEMMs, gt and B_3 e materialised, separate by-source passes).  Same arithmetic up to
    the order of the fp32 sums: logits within 2e-6 (rel-L2), gradients at the bar they have against the oracle, every
    configuration run-to-run bit-identical; on generator node ids and on shuffled ones (the renumbered index)."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine
    dev = _dev()
    for ids in ("sorted", "shuffled"):
        model, src, dst, n, inp = _model_and_inputs(30000, 256, 3, 21, dev)
        pe_np = inp["pe"]
        if ids == "shuffled":
            p = np.random.default_rng(6).permutation(n).astype(np.int32)
            src, dst = p[src], p[dst]
            pe_s = np.empty_like(pe_np)
            pe_s[p] = pe_np
            pe_np = pe_s
        g = G.AssemblyGraph(src, dst, n).to(dev)
        e, pe, y = torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(pe_np).to(dev), torch.from_numpy(inp["y"]).to(dev)
        crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))

        def run(**opts):
            with engine.options(**opts):
                model.zero_grad(set_to_none=True)
                s = model(g, None, e, pe)
                loss = crit(s.squeeze(-1), y)
                loss.backward()
                torch.cuda.synchronize()
                return s.detach().clone(), loss.item(), {k: v.grad.clone() for k, v in model.named_parameters()}
        base = run(WIDE_FUSED=False, TWO_SIDED=False, TWO_SIDED_FWD=False)
        for name, opts in (("fused kernels", dict(TWO_SIDED=False, TWO_SIDED_FWD=False)),
                           ("fused kernels + two-sided sweeps (default)", dict())):
            s1, l1, g1 = run(**opts)
            s2, l2, g2 = run(**opts)
            assert torch.equal(s1, s2) and l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1), (ids, name, "not deterministic")
            r = rel_l2(s1.cpu().numpy(), base[0].cpu().numpy())
            print(f"H = 256, {name} vs the generic route [{ids}]: logits rel_l2 = {r:.2e}")
            assert r <= 2e-6 and abs(l1 - base[1]) <= 1e-6 * abs(base[1]), (ids, name, r)
            gmax = max(float(v.abs().max()) for v in base[2].values())
            bad = []
            for k in base[2]:
                a, b = g1[k].double(), base[2][k].double()
                rr = float((a - b).norm() / b.norm().clamp_min(1e-30))
                if rr > GRAD_L2 and float((a - b).abs().max()) > 1e-6 * gmax:
                    bad.append((k, rr))
            assert not bad, (ids, name, bad)


@pytest.mark.mode_independent
def test_device_built_sweep_plan_equals_the_host_built_one():
    """gnm_graph_build_sweep_plan_device (one wave per sweep workgroup, the slot allocator on a stack in LDS) against the host
    builder on the same index: plan words bit for bit, the same unserved nodes, the same peak slot count -- for the synthetic
    assembly graph with generator and shuffled (renumbered) node ids, both partitions, and the graphs without a band."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import graph as gmod, synth
    dev = _dev()
    cases = {}
    src, dst, n = synth.make_graph(60000, 5)
    cases["assembly graph"] = (src, dst, n)
    p = np.random.default_rng(9).permutation(n).astype(np.int32)
    cases["assembly graph, shuffled ids"] = (p[src], p[dst], n)
    cases.update(_adversarial_graphs())
    for name, (src, dst, n) in cases.items():
        g = G.AssemblyGraph(src, dst, n).to(dev)
        idx = g.index(dev)
        for wg in (1, 2):
            host = g.sweep_plan(dev, wg)
            devp = gmod.build_sweep_plan_device(idx, n, dev, wg)
            torch.cuda.synchronize()
            assert devp["nodes_per_block"] == host["nodes_per_block"]
            assert torch.equal(devp["sinfo"], host["sinfo"]), (name, wg, "sinfo")
            assert torch.equal(devp["dinfo"], host["dinfo"]), (name, wg, "dinfo")
            fix = devp["fix_nodes"]
            assert torch.equal(fix[fix >= 0], host["fix_nodes"]) and bool((fix[fix >= 0] == torch.nonzero(fix >= 0).squeeze(1)).all()), (name, wg, "fix list")
            assert int(devp["peak_dev"].item()) == host["peak_live"], (name, wg, "peak")


@pytest.mark.default_mode_only
def test_device_born_graph_runs_the_two_sided_sweeps():
    """A graph built from DEVICE tensors (what cluster.induced_subgraph returns for a mini-batch) gets its plan on the device and runs
    the two-sided sweeps: logits and gradients equal those of the same graph built on the host (same plan -> same arithmetic,
    bit for bit), and GNM_DEVICE_PLANS=0 / graph.DEVICE_PLANS = False (the separate passes) agrees to the usual bars."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import graph as gmod
    dev = _dev()
    model, src, dst, n, inp = _model_and_inputs(30000, 128, 3, 4, dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))

    def run(g):
        model.zero_grad(set_to_none=True)
        s = model(g, None, e, pe)
        crit(s.squeeze(-1), y).backward()
        torch.cuda.synchronize()
        return s.detach().clone(), {k: v.grad.clone() for k, v in model.named_parameters()}
    g_host = G.AssemblyGraph(src, dst, n, node_order="keep").to(dev)
    g_dev = G.AssemblyGraph.from_tensors(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
    assert g_dev.sweep_plan(dev, 1) is not None and g_dev.sweep_plan(dev, 2) is not None
    s0, g0 = run(g_host)
    s1, g1 = run(g_dev)
    assert torch.equal(s0, s1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    old = gmod.DEVICE_PLANS
    try:
        gmod.DEVICE_PLANS = False
        g_sep = G.AssemblyGraph.from_tensors(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
        assert g_sep.sweep_plan(dev, 1) is None
        s2, g2 = run(g_sep)
    finally:
        gmod.DEVICE_PLANS = old
    assert rel_l2(s2.cpu().numpy(), s0.cpu().numpy()) <= 2e-6
    gmax = max(float(v.abs().max()) for v in g0.values())
    bad = [(k, float((g2[k].double() - g0[k].double()).norm() / g0[k].double().norm().clamp_min(1e-30))) for k in g0
           if float((g2[k] - g0[k]).abs().max()) > 1e-6 * gmax]
    assert all(r <= GRAD_L2 for _, r in bad), bad


def _adversarial_graphs():
    """Graphs the sweep plan has to get right without the band it was designed for."""
    rng = np.random.default_rng(12)
    out = {}
    n, e = 3000, 20000
    out["random (no locality at all)"] = (rng.integers(0, n, e).astype(np.int32), rng.integers(0, n, e).astype(np.int32), n)
    hub = np.concatenate([np.full(6000, 7), rng.integers(0, 2000, 3000)]).astype(np.int32)       # one destination with 6000 in-edges
    out["hub destination (6000 in-edges = 375 tiles)"] = (rng.integers(0, 2000, 9000).astype(np.int32), hub, 2000)
    out["hub source (5000 out-edges)"] = (np.concatenate([np.full(5000, 11), rng.integers(0, 1500, 2000)]).astype(np.int32),
                                          rng.integers(0, 1500, 7000).astype(np.int32), 1500)
    path = np.arange(0, 4999, dtype=np.int32)
    out["path (one in-edge, one out-edge per node)"] = (path, path + 1, 5000)
    out["path reversed + self loops + duplicates"] = (np.concatenate([path + 1, path[:50], path[:50]]), np.concatenate([path, path[:50], path[:50]]), 5000)
    k = 40                                                                                              # more open sources than slots
    s_ = np.repeat(np.arange(k, dtype=np.int32), 30)
    d_ = (100 + np.tile(np.arange(30, dtype=np.int32), k) * 3 + np.repeat(np.arange(k, dtype=np.int32) % 3, 30)).astype(np.int32)
    out["40 sources interleaved over the same destinations (slot overflow)"] = (s_, d_, 400)
    out["tiny"] = (np.array([0, 1, 2, 2], np.int32), np.array([1, 2, 0, 2], np.int32), 5)
    return out


@pytest.mark.default_mode_only
@pytest.mark.parametrize("ids", ["generator", "shuffled"])
def test_layernorm_two_sided_backward_sweep_matches_the_separate_passes(ids):
    """Round 6, batch_norm=False at H = 128: gnm_ln_edge_bwd_top + gnm_ln_edge_bwd_src_fix (the LayerNorm form of the two-sided top
    sweep: by-destination AND by-source sums of one pass, gt complete inside its row) against gnm_ln_edge_bwd_dst + gnm_ln_edge_bwd_src.
    The forward is the same under both switches, so the gradients differ by summation order only: every tensor within 1e-5 rel-L2 (or
    under the absolute floor), two runs of the sweep bit-identical.  Also with a zero-padded width (96 -> 128: dead channels stay out of
    the row statistics inside the sweep too)."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine, synth
    dev = _dev()
    for H in (128, 96):
        L, seed = 3, 13
        src, dst, n = synth.make_graph(16000, seed)
        inp = synth.make_inputs(src, dst, n, seed)
        pe_np = inp["pe"]
        if ids == "shuffled":
            p = np.random.default_rng(5).permutation(n).astype(np.int32)
            src, dst = p[src], p[dst]
            pe_s = np.empty_like(pe_np)
            pe_s[p] = pe_np
            pe_np = pe_s
        model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, False, 16)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, seed).items()})
        model.to(dev)
        g = G.AssemblyGraph(src, dst, n).to(dev)
        e, pe, y = torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(pe_np).to(dev), torch.from_numpy(inp["y"]).to(dev)
        crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
        assert g.sweep_plan(dev, 1) is not None

        def run(sweep, chain=True):
            with engine.options(LN_SWEEP=sweep, CHAIN=chain):
                model.zero_grad(set_to_none=True)
                s = model(g, None, e, pe)
                loss = crit(s.squeeze(-1), y)
                loss.backward()
                torch.cuda.synchronize()
                return s.detach().clone(), {k: v.grad.clone() for k, v in model.named_parameters()}
        s0, g0 = run(False)
        s1, g1 = run(True, chain=False)         # the sweep, layer by layer (gnm_ln_edge_bwd_top + gnm_edge_bwd_fused_gt)
        s2, g2 = run(True, chain=False)
        s3, g3 = run(True)                      # the chained schedule (gnm_ln_edge_bwd_chain: fused(i) with gt given + sweep(i-1))
        s4, g4 = run(True)
        assert torch.equal(s0, s1) and torch.equal(s1, s2) and torch.equal(s2, s3), "the forward does not depend on the backward's schedule"
        assert all(torch.equal(g1[k], g2[k]) for k in g1), "the sweep is not run-to-run deterministic"
        assert all(torch.equal(g3[k], g4[k]) for k in g3), "the chained LayerNorm backward is not run-to-run deterministic"
        gmax = max(float(v.abs().max()) for v in g0.values())
        for what, gx in (("sweep", g1), ("chained", g3)):
            worst = 0.0
            for k in g0:
                a, b = gx[k].double(), g0[k].double()
                rr = float((a - b).norm() / b.norm().clamp_min(1e-30))
                ok = rr <= 1e-5 or float((a - b).abs().max()) <= 1e-6 * gmax
                worst = max(worst, rr if float(b.norm()) > 1e-6 * gmax else 0.0)
                assert ok, (what, H, k, rr, float((a - b).abs().max()), gmax)
            print(f"LayerNorm backward, {what} vs separate passes [{ids}, H = {H}]: worst gradient rel_l2 = {worst:.2e}")


@pytest.mark.default_mode_only
@pytest.mark.parametrize("H", [128, 256])
def test_two_sided_sweeps_on_graphs_without_a_band(H):
    """(H = 256: the same sweeps once per 128-column half, row pitch 256.)
    The two-sided sweeps on graphs the plan was not designed for (no locality, hubs with thousands of rows, paths, self loops and
    duplicates, more simultaneously open sources than accumulator slots, five nodes): whatever the plan leaves to the fix-up
    kernels, logits and gradients equal the separate-pass schedule's (logits 2e-6 rel-L2, gradients at the oracle bar) and two runs
    are bit-identical."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine, synth
    dev = _dev()
    L = 2
    sd = synth.synth_state_dict(H, L, 3)
    for name, (src, dst, n) in _adversarial_graphs().items():
        rng = np.random.default_rng(len(name))
        E = src.size
        model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.to(dev)
        g = G.AssemblyGraph(src, dst, n).to(dev)
        e = torch.from_numpy(rng.standard_normal((E, 2)).astype(np.float32)).to(dev)
        pe = torch.from_numpy(rng.standard_normal((n, 18)).astype(np.float32)).to(dev)
        y = torch.from_numpy((rng.random(E) < 0.7).astype(np.float32)).to(dev)
        crit = G.BCEWithLogitsLoss(0.4)
        p1, p2 = g.sweep_plan(dev, 1), g.sweep_plan(dev, 2)

        def run(fwd2, bwd2):
            with engine.options(TWO_SIDED=bwd2, TWO_SIDED_FWD=fwd2):
                model.zero_grad(set_to_none=True)
                s = model(g, None, e, pe)
                loss = crit(s.squeeze(-1), y)
                loss.backward()
                torch.cuda.synchronize()
                return s.detach().clone(), {k: v.grad.clone() for k, v in model.named_parameters()}
        s0, g0 = run(False, False)
        s1, g1 = run(True, False)          # the forward sweep alone: the logits move by a summation order
        s2, g2 = run(True, True)           # + the backward sweep on the SAME forward: the gradients move by a summation order
        s3, g3 = run(True, True)
        assert bool(torch.isfinite(s2).all()) and all(bool(torch.isfinite(v).all()) for v in g2.values()), name
        assert torch.equal(s1, s2) and torch.equal(s2, s3) and all(torch.equal(g2[k], g3[k]) for k in g2), name
        r = rel_l2(s1.cpu().numpy(), s0.cpu().numpy())
        gmax = max(float(v.abs().max()) for v in g1.values())
        rg = max(float((g2[k].double() - g1[k].double()).norm() / g1[k].double().norm().clamp_min(1e-30)) for k in g1
                 if float((g2[k] - g1[k]).abs().max()) > 1e-6 * gmax) if any(
                     float((g2[k] - g1[k]).abs().max()) > 1e-6 * gmax for k in g1) else 0.0
        print(f"{name}: N={n} E={E} fix-up nodes bwd {p1['nfix']} / fwd {p2['nfix']}, peak slots {p1['peak_live']}; logits rel_l2 {r:.1e}, "
              f"gradients (backward sweep vs separate pass, same forward) worst rel_l2 {rg:.1e}")
        assert r <= 2e-6, (name, r)
        assert rg <= 2e-5, (name, rg)


@pytest.mark.default_mode_only
def test_chr1_scale_inference_at_size():
    """BASELINE config 5 at its size (chr1 = 4.03 x chr19 -> R=3 M reads, N=6 M nodes, E~30 M edges,
    H=128, L=8), forward only under no_grad as inference.py:444-454 calls the model.  E*H = 3.9 G elements
    exceeds 2^31, so this is the case that guards every row-offset computation in the kernels.  Checks:
    finite; a repeated run is bit-identical; peak memory stays below 8 [E,H] units (nothing is kept for a
    backward); relabelling the edges (reversed edge-id order) permutes the logits and nothing else."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import features, synth
    dev = _dev()
    R, H, L = 3_000_000, 128, 8
    src, dst, n = synth.make_graph(R, seed=5)
    E = int(src.size)
    assert E * H > 2 ** 31
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0).items()})
    model.to(dev).eval()
    g = G.AssemblyGraph(src, dst, n).to(dev)
    pe = features.positional_encoding(g)                       # degrees + PageRank PE on the device (utils.py:97-138)
    gen = torch.Generator(device=dev).manual_seed(5)
    e = torch.randn(E, 2, device=dev, generator=gen)
    g.index()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        s0 = model(g, None, e, pe)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        s1 = model(g, None, e, pe)
    unit = 4.0 * E * H
    print(f"chr1-scale inference: N={n} E={E} peak {peak / 2**30:.1f} GiB = {peak / unit:.2f} [E,H] units")
    assert s0.shape == (E, 1) and bool(torch.isfinite(s0).all()) and torch.equal(s0, s1)
    assert float(s0.std()) > 1e-3                               # not a constant
    assert peak < 8 * unit
    del s1
    with torch.no_grad():
        g2 = G.AssemblyGraph(src[::-1].copy(), dst[::-1].copy(), n).to(dev)
        s2 = model(g2, None, e.flip(0).contiguous(), pe)
    want = s0.flip(0)
    r = float((s2 - want).double().norm() / want.double().norm())
    print(f"chr1-scale edge-id reversal: rel_l2 {r:.2e}, max abs {float((s2 - want).abs().max()):.2e}")
    assert r <= 2e-5 and float((s2 - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-5


@pytest.mark.mode_independent
def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the driver's keys, the roofline and cpu_baseline objects."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--reads", "30000", "--steps", "2", "--warmup", "1",
                          "--cpu-reads", "2000"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "edges/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32 (f16x2 split products, f32 accumulate)" and r["config"]["matmul"] == "f16x2"
    assert r["data"] == "synthetic" and "workload" in r["config"]
    assert abs(r["value"] - r["config"]["edges"] / (r["ms_per_step"] / 1e3)) <= 1e-6 * r["value"]
    rf = r["roofline"]                 # the step against the HBM roofline, per-kernel table inside
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and "traffic" in rf and "traffic_source" in rf
    alg = 32 * r["config"]["hidden"] * r["config"]["layers"] * r["config"]["edges"]
    assert abs(rf["achieved"] * 1e9 - alg / (r["ms_per_step"] / 1e3)) <= 1e-6 * rf["achieved"] * 1e9
    ks = rf["kernels"]
    assert ks and rf["dominant_kernel"] == ks[0] and all(k["share_of_step"] >= 0.02 for k in ks)
    ops = {k["op"] for k in ks}
    assert "gnm_node_proj_bwd" not in ops            # its two kernels are timed on their own
    for k in ks:
        if "hbm" in k:
            assert 0 < k["hbm"]["frac"] < 1.0
        if "mfma" in k:
            assert 0 < k["mfma"]["frac"] < 1.0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "edges/s" and cb["sample"]
    assert [a["matmul"] for a in r["alt_matmul"]] == ["bf16x3", "f32"] and all(a["value"] > 0 for a in r["alt_matmul"])
    assert "1/" in cb["sample"]                      # the sample states its ratio to the GPU workload


def test_side_stream_schedule_and_per_call_caps_change_nothing_but_rounding():
    """The chained backward launches the node weight-gradient kernel on a side stream beside the by-source pass
    (engine.TN_SIDE), in one launch or in two sized to the HBM-bound windows of an iteration (engine.TN_SPLIT); the
    workgroups-per-CU caps of the two kernels are ARGUMENTS of their launches (gnm.h max_blocks_per_cu), no process-wide state.  Same kernels: another cap only changes the number of per-workgroup
    partial slabs, i.e. the summation order -> gradients equal to fp32 round-off, and every setting is deterministic."""
    from gnnome_assembly_amd import engine
    dev = _dev()

    def run():
        model, src, dst, n, inp = _model_and_inputs(20000, 128, 3, 2, dev)
        import gnnome_assembly_amd as G
        g = G.AssemblyGraph(src, dst, n).to(dev)
        crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
        s = model(g, None, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev))
        crit(s.squeeze(-1), torch.from_numpy(inp["y"]).to(dev)).backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in model.named_parameters()}
    base = run()
    for setting in ((False, 0, 0, "next", True), (True, 1, 2, "next", False), (True, 0, 4, "next", True), (True, 2, 8, "now", True),
                    (True, 0, 4, "now", False)):
        with engine.options(TN_SIDE=setting[0], TN_SIDE_CAP=setting[1], SRC_SIDE_CAP=setting[2], TN_AT=setting[3],
                            TN_SPLIT=setting[4]):
            co, co2 = run(), run()
        for k in base:
            assert torch.equal(co[k], co2[k]), (setting, k)                    # still deterministic
            d = float((co[k] - base[k]).abs().max())
            assert d <= 1e-5 * float(base[k].abs().max()) + 1e-9, (setting, k, d)


@pytest.mark.mode_independent
@pytest.mark.parametrize("prog", ["host_layer", "host_step"])
def test_cxx_host_through_the_c_abi(tmp_path, prog):
    """tests/cabi/*.cpp: C++ programs with no Python and no torch that run on hipMalloc'd buffers through include/gnm.h +
    libgnm.so and check themselves against their own fp64 loops.  host_layer: one layer forward on the round-1 entry points
    (separate gate / by-source passes).  host_step: the path bench.py measures -- two layers forward through the sweep plans and
    the two-sided gate kernel, backward through the top sweep, the chained edge kernel and the fused node-side kernels
    (gnm_tn128_bgrad, gnm_node_proj_bwd_nn_stats), every gradient against fp64 loops on the device's relu branches."""
    import shutil
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = shutil.which("hipcc")
    assert hipcc, "hipcc not found"
    exe = str(tmp_path / prog)
    libdir = os.path.join(repo, "gnnome_assembly_amd")
    cc = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(repo, "include"),
                         os.path.join(repo, "tests", "cabi", prog + ".cpp"), "-L", libdir, "-lgnm", "-o", exe],
                        capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", "")))
    print(run.stdout)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"cabi_{prog}.txt"), "w") as f:
        f.write(run.stdout)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr


def test_degenerate_constant_features_stay_finite_and_match_the_oracle():
    """Every node and edge carries the same features: BatchNorm sees zero variance in every channel
    (rstd = 1/sqrt(eps)); nothing may blow up and the logits still agree with the fp64 oracle."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    src, dst, n = synth.make_graph(300, 4, permute_edge_ids=True)
    deg_regular = np.ones((n, 18), np.float32) * 0.25
    e_raw = np.ones((src.size, 2), np.float32) * np.array([0.5, -1.0], np.float32)
    H, L = 128, 3
    sd = synth.synth_state_dict(H, L, 2)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    with torch.no_grad():
        got = model(g, None, torch.from_numpy(e_raw).to(dev), torch.from_numpy(deg_regular).to(dev))
        want = orc.model_forward(sd_to_torch(sd, torch.float64), torch.from_numpy(src).long(), torch.from_numpy(dst).long(), n,
                                 torch.from_numpy(e_raw).double(), torch.from_numpy(deg_regular).double(), True)
        want32 = orc.model_forward(sd_to_torch(sd, torch.float32), torch.from_numpy(src).long(), torch.from_numpy(dst).long(), n,
                                   torch.from_numpy(e_raw), torch.from_numpy(deg_regular), True)
    assert bool(torch.isfinite(got).all())
    # With zero variance the normalised value is a rounding residual times 316 (rstd = 1/sqrt(eps)): ANY fp32 evaluation is
    # far from the fp64 one here, the reference's own arithmetic included.  The bar is therefore the usual 1e-4 of the output
    # scale OR three times the distance of the oracle's fp32 run from its fp64 run -- not a hand-picked 2e-3.
    scale = max(1.0, float(want.abs().max()))
    d = float((got.cpu().double() - want).abs().max())
    d32 = float((want32.double() - want).abs().max())
    print(f"constant-feature graph: max |logit - oracle64| = {d:.2e} (oracle fp32 vs fp64: {d32:.2e}), |logit| up to {scale:.3f}")
    assert d <= max(1e-4 * scale, 3.0 * d32)


def test_non_finite_inputs_stay_confined_and_the_split_mode_divergence_is_as_documented():
    """An inf operand of a split-mode (bf16x3) product gives NaN where the reference's fp32 arithmetic gives +-inf:
    x = hi + mid + lo with hi = inf leaves mid = inf - inf = NaN, and even a clean split would meet inf * w_hi +
    inf * w_mid with opposite signs.  The fp32-MFMA mode propagates inf like torch.  Either way the damage is
    CONFINED to the rows that hold the non-finite value (no kernel mixes rows of a [M,K] operand in the forward
    product), NaN stays NaN in both modes, and all other rows are bit-identical to the run without it.
    INTEGRATION.md ("Numerics") tells callers which mode to pick if they rely on inf semantics."""
    from gnnome_assembly_amd import _lib, engine
    dev = _dev()
    mode = _lib.get_matmul_mode()
    rng = np.random.default_rng(5)
    M, N, K = 4096, 128, 128                                  # the shape class of every H = 128 layer product
    X = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev)
    Wt = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32) + 0.01).to(dev)
    clean = engine.gemm(engine.NT, X, Wt, torch.empty(M, N, device=dev))
    Xi = X.clone()
    Xi[7, 3] = float("inf")
    Xi[100, 5] = float("nan")
    out = engine.gemm(engine.NT, Xi, Wt, torch.empty(M, N, device=dev))
    torch.cuda.synchronize()
    rows = torch.ones(M, dtype=torch.bool, device=dev)
    rows[7] = rows[100] = False
    assert torch.equal(out[rows], clean[rows])                # confined to the two rows
    assert bool(torch.isnan(out[100]).all())                  # NaN in -> NaN out, both modes
    assert not bool(torch.isfinite(out[7]).any())             # the inf row is non-finite everywhere ...
    ref = Xi[7:8].cpu() @ Wt.cpu().t()                        # ... torch fp32: +-inf by the sign of W[:, 3]
    assert bool(torch.isinf(ref).all())
    if mode == "f32":
        assert torch.equal(out[7:8].cpu(), ref)               # the fp32-MFMA mode keeps the reference's inf semantics
    else:
        assert bool(torch.isnan(out[7]).any())                # bf16x3: the documented divergence


@pytest.mark.parametrize("name", ["in48_out32_bn", "in32_out32_bn_nores", "in128_out128_bn_nores", "in16_out32_ln",
                                  "in32_out32_ln_nores"])
def test_standalone_layer_variants_match_the_reference_layer(name):
    """GatedGCN_1d(residual=False) and in_channels != out_channels (the reference drops the residual then:
    gated_gcn_full.py:41-42,124-125,151-152) -- arguments of the class whose signature is the boundary, never used by
    the model -- against outputs and gradients of the REFERENCE's own layer run in fp64
    (tests/golden/make_golden_layer.py -> layer_variants.npz)."""
    import gnnome_assembly_amd as G
    from helpers import GOLDEN, LAYER_VARIANTS, layer_variant_case
    dev = _dev()
    z = np.load(os.path.join(GOLDEN, "layer_variants.npz"))
    cin, cout, bn, res = LAYER_VARIANTS[name]
    c = layer_variant_case(name)
    layer = G.layers.GatedGCN_1d(cin, cout, bn, residual=res)
    assert layer.residual == bool(z[f"{name}/residual_used"])
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in c["sd"].items()}, strict=True)
    layer.to(dev)
    graph = G.AssemblyGraph(c["src"], c["dst"], c["n"]).to(dev)
    h = torch.from_numpy(c["h0"]).to(dev).requires_grad_(True)
    e = torch.from_numpy(c["e0"]).to(dev).requires_grad_(True)
    h1, e1 = layer(graph, h, e)
    assert h1.shape == (c["n"], cout) and e1.shape == (c["src"].size, cout)
    ((h1 * torch.from_numpy(c["wh"]).to(dev)).sum() + (e1 * torch.from_numpy(c["we"]).to(dev)).sum()).backward()
    torch.cuda.synchronize()
    rows = []
    got = {"h1": h1, "e1": e1, "gh": h.grad, "ge": e.grad, **{"g/" + k: p.grad for k, p in layer.named_parameters()}}
    for k, v in got.items():
        _cmp(k, v, z[f"{name}/{k}"], rows)
    _report(rows, f"layer_variant_{name}.txt")
    gmax = max(r[3] for r in rows if r[0].startswith("g"))
    # biases in front of a BatchNorm have analytically zero gradients (the reference's are round-off as well)
    bad = [r for r in rows if r[1] > (2e-5 if r[0] in ("h1", "e1") else GRAD_L2) and r[2] > max(GRAD_ABS_FLOOR, 1e-6 * gmax)]
    assert not bad, bad
