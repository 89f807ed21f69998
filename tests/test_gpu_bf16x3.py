"""The "bf16x3" matmul mode of the fused kernels (include/gnm.h: gnm_set_matmul_mode(1), the library
default): every fp32 operand is split exactly into three bf16 terms and each product is formed from six
bf16 MFMAs with fp32 accumulation.  The claim to verify is that this is an fp32-class matmul.  The whole
of tests/test_gpu_parity.py runs under both modes (its autouse `matmul_mode` fixture); this file holds
the mode-vs-mode comparisons: per-kernel error against fp64 next to the fp32-MFMA mode's, logits no
further from the reference's fp64 run than 3x the reference's own fp32 run, and a 40-step training
trajectory that follows the fp32-MFMA mode's."""
import numpy as np
import pytest
import torch

import test_gpu_parity as base
from helpers import load_case, rel_l2, sd_to_torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def bf16x3_mode():
    from gnnome_assembly_amd import _lib
    _lib.set_matmul_mode("bf16x3")
    assert _lib.get_matmul_mode() == "bf16x3"
    yield
    _lib.set_matmul_mode(_lib.DEFAULT_MATMUL_MODE)


def test_mode_switch_is_validated():
    from gnnome_assembly_amd import _lib
    with pytest.raises(_lib.GnmError):
        _lib.set_matmul_mode("bf16")
    lib = _lib.load()
    assert lib.gnm_set_matmul_mode(7) != 0 and lib.gnm_get_matmul_mode() == 1


@pytest.mark.parametrize("M", [64, 1000, 4097, 70001])
def test_fused_kernels_match_fp64_like_the_fp32_mode(M):
    """edge_t_fused / node_proj fwd+bwd / edge_bwd_fused on random data, ragged M: error against an
    fp64 evaluation must be fp32 round-off, and no worse than twice the default mode's error."""
    from gnnome_assembly_amd import _lib, engine
    dev = base._dev()
    H = 128
    gen = torch.Generator(device=dev).manual_seed(M)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
    x, W5, b5 = rnd(M, H), rnd(5 * H, H) / H ** 0.5, rnd(5 * H)
    gP, gh = rnd(M, 5 * H), rnd(M, H)

    def run():
        lib, sc, st = _lib.load(), engine.scratch(dev), engine._stream()
        ptr = engine._ptr
        P = torch.empty(M, 5 * H, device=dev)
        need = lib.gnm_rowtile_workspace_bytes(5 * H)
        ws = sc.ws(need)
        engine._call("gnm_node_proj_fwd", M, H, 5 * H, ptr(x), ptr(W5), ptr(b5), ptr(P), ptr(ws), need, st)
        gh_in, gW, gb = torch.empty(M, H, device=dev), torch.empty(5 * H, H, device=dev), torch.empty(5 * H, device=dev)
        need = lib.gnm_node_proj_bwd_workspace_bytes(5 * H)
        ws = sc.ws(need)
        engine._call("gnm_node_proj_bwd", M, H, 5 * H, ptr(gP), ptr(x), ptr(W5), ptr(gh), ptr(gh_in), ptr(gW), ptr(gb),
                     ptr(sc.partials), ptr(ws), need, st)
        torch.cuda.synchronize()
        return [P, gh_in, gW, gb]
    want = [x.double() @ W5.double().T + b5.double(), gh.double() + gP.double() @ W5.double(),
            gP.double().T @ x.double(), gP.double().sum(0)]
    got_b3 = run()
    _lib.set_matmul_mode("f32")
    got_f32 = run()
    _lib.set_matmul_mode("bf16x3")
    for name, b3, f32, w in zip(["P", "gh_in", "gW5", "gb5"], got_b3, got_f32, want):
        rb, rf = rel_l2(b3.cpu().numpy(), w.cpu().numpy()), rel_l2(f32.cpu().numpy(), w.cpu().numpy())
        print(f"M={M} {name}: rel_l2 bf16x3={rb:.2e} f32={rf:.2e}")
        assert rb <= 2e-6 and rb <= 2 * rf + 2e-7, (name, rb, rf)


@pytest.mark.parametrize("fname", ["tiny_h128l8_s0.npz", "small_h128l8_s0.npz", "small_h128l8_s1.npz"])
def test_logits_and_loss_match_golden(fname):
    """Logits and loss bar of test_gpu_parity.test_model_matches_golden, plus: no further from the reference's
    fp64 run than 3x the reference's own fp32 run is."""
    dev = base._dev()
    z, sd, H, L, bn = load_case(fname)
    model, graph, x, e, pe, y, crit = base._run_model(z, sd, H, L, dev)
    scores = model(graph, x, e, pe)
    loss = crit(scores.squeeze(-1), y)
    loss.backward()
    torch.cuda.synchronize()
    base.assert_parity(scores.detach().cpu().numpy(), z["scores64"], f"{fname} logits (bf16x3) vs reference fp64")
    ours, ref32 = rel_l2(scores.detach().cpu().numpy(), z["scores64"]), rel_l2(z["scores32"], z["scores64"])
    print(f"{fname}: logits rel_l2 bf16x3={ours:.2e} reference-fp32={ref32:.2e}")
    assert ours <= 3 * ref32
    assert abs(loss.item() - float(z["loss64"])) <= 1e-5 * max(1.0, abs(float(z["loss64"])))
    for k, prm in model.named_parameters():
        assert bool(torch.isfinite(prm.grad).all()), k


def test_training_trajectory_matches_the_fp32_mode():
    """40 Adam steps of the 8-layer model on one graph from the same initialisation: the loss sequence of
    the split mode follows the default mode's to 1e-4 relative (measured 1.1e-5) -- the two modes are
    interchangeable for training, not only for one forward pass."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import _lib, synth
    dev = base._dev()
    src, dst, n = synth.make_graph(20000, 3, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, 3)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))

    def run(mode, steps=40):
        _lib.set_matmul_mode(mode)
        model = G.GraphGatedGCNModel(1, 2, 128, 16, 8, 64, True, 16)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(128, 8, 1).items()})
        model.to(dev)
        crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        out = []
        for _ in range(steps):
            opt.zero_grad()
            loss = crit(model(g, None, e, pe).squeeze(-1), y)
            loss.backward()
            opt.step()
            out.append(loss.item())
        return np.array(out)
    split = run("bf16x3")
    exact = run("f32")
    _lib.set_matmul_mode("bf16x3")
    d = float(np.max(np.abs(split - exact) / exact))
    print(f"loss trajectories over 40 steps: max relative difference {d:.2e}; last {exact[-1]:.6f} vs {split[-1]:.6f}")
    assert np.all(np.isfinite(split)) and d <= 1e-4
