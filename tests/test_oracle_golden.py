"""CPU: pin the oracle (oracle/gatedgcn_oracle.py) against the golden vectors produced by
the reference's own code (tests/golden/make_golden.py), and check the hand-derived backward
used as the per-kernel checker against autograd."""
import numpy as np
import pytest
import torch

from conftest import golden_files
from helpers import load_case, grad_stride_of, sd_to_torch, rel_l2
from oracle import gatedgcn_oracle as orc


def _inputs(z, dtype):
    t = lambda k: torch.from_numpy(z[k]).to(dtype)  # noqa: E731
    return (torch.from_numpy(z["src"]).long(), torch.from_numpy(z["dst"]).long(), int(z["n"]),
            t("e_raw"), t("pe"), t("y"), float(z["pos_weight"]))


@pytest.mark.parametrize("fname", golden_files())
def test_forward_matches_reference(fname):
    z, sd, H, L, bn = load_case(fname)
    for dtype, tag, tol in ((torch.float64, "64", 1e-12), (torch.float32, "32", 2e-5)):
        src, dst, n, e_raw, pe, y, pw = _inputs(z, dtype)
        p = sd_to_torch(sd, dtype)
        with torch.no_grad():
            s, layers = orc.model_forward(p, src, dst, n, e_raw, pe, bn, return_layers=True)
            loss = orc.bce_loss(s, y, pw)
        assert s.shape == (src.numel(), 1)
        assert rel_l2(s.numpy(), z["scores" + tag]) < tol
        assert abs(loss.item() - float(z["loss" + tag])) < max(tol, 1e-6 if tag == "32" else 0) * 10
        if tag == "64" and L <= 2:
            rs = int(z["row_stride"])
            for i, (h, e) in enumerate(layers):
                assert rel_l2(h.numpy()[::rs], z[f"layer{i}/h"]) < 1e-12
                assert rel_l2(e.numpy()[::rs], z[f"layer{i}/e"]) < 1e-12


@pytest.mark.parametrize("fname", golden_files())
def test_autograd_grads_and_adam_match_reference(fname):
    z, sd, H, L, bn = load_case(fname)
    src, dst, n, e_raw, pe, y, pw = _inputs(z, torch.float64)
    p = sd_to_torch(sd, torch.float64, requires_grad=True)
    loss = orc.bce_loss(orc.model_forward(p, src, dst, n, e_raw, pe, bn), y, pw)
    loss.backward()
    stride = grad_stride_of(z, H)
    grads = {k: v.grad for k, v in p.items()}
    for k in p:
        got = grads[k].numpy().reshape(-1)[::stride]
        want = z["grad/" + k]
        assert np.abs(got - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), k
    new = orc.adam_step({k: v.detach() for k, v in p.items()}, grads)
    for k in p:
        got = new[k].numpy().reshape(-1)[::stride]
        assert np.abs(got - z["adam/" + k]).max() < 1e-9, k


@pytest.mark.parametrize("fname", golden_files("h64l1") + golden_files("small_h128l8_s0"))
def test_manual_backward_matches_autograd(fname):
    z, sd, H, L, bn = load_case(fname)
    src, dst, n, e_raw, pe, y, pw = _inputs(z, torch.float64)
    p = sd_to_torch(sd, torch.float64, requires_grad=True)
    loss = orc.bce_loss(orc.model_forward(p, src, dst, n, e_raw, pe, bn), y, pw)
    loss.backward()
    with torch.no_grad():
        s, l2, g = orc.manual_forward_backward({k: v.detach() for k, v in p.items()},
                                               src, dst, n, e_raw, pe, y, pw)
    assert abs(l2.item() - loss.item()) < 1e-13
    assert rel_l2(s.numpy(), z["scores64"]) < 1e-12
    for k in p:
        a, b = g[k].numpy(), p[k].grad.numpy()
        assert a.shape == b.shape, k
        # biases in front of a BatchNorm have an exactly-zero gradient (pure round-off)
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()) + 1e-15, k


@pytest.mark.parametrize("fname", golden_files("tiny_h64l1"))
def test_harness_pins(fname):
    """train.py:181 ratio, utils.calculate_tfpn counts, 3-step Adam loss sequence, eval==train."""
    z, sd, H, L, bn = load_case(fname)
    src, dst, n, e_raw, pe, y, pw = _inputs(z, torch.float64)
    assert bool(z["eval_equals_train"])
    assert abs(float((y == 1).sum() / (y == 0).sum()) - float(z["pos_to_neg_ratio"])) < 1e-6
    p = {k: v.requires_grad_(True) for k, v in sd_to_torch(sd, torch.float64).items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3)
    seq = []
    for step in range(3):
        s = orc.model_forward(p, src, dst, n, e_raw, pe, bn)
        loss = orc.bce_loss(s, y, pw)
        if step == 0:
            pred = torch.round(torch.sigmoid(s.reshape(-1)))
            tfpn = [int(((pred == a) & (y == b)).sum()) for a, b in ((1, 1), (0, 0), (1, 0), (0, 1))]
            assert tfpn == list(z["tfpn"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        seq.append(loss.item())
    assert np.abs(np.array(seq) - z["loss_seq64"]).max() < 1e-10


def test_pagerank_pe_matches_reference(golden_dir):
    import os
    from gnnome_assembly_amd import synth
    z = np.load(os.path.join(golden_dir, "pe_pagerank.npz"))
    pe = synth.pagerank_pe(z["src"], z["dst"], int(z["n"]))
    assert np.abs(pe - z["pe"]).max() <= 1e-7 * np.abs(z["pe"]).max()
    assert np.array_equal(np.bincount(z["dst"], minlength=int(z["n"])), z["in_deg"].astype(np.int64))


def test_oracle_layer_variants_match_the_reference_layer():
    """GatedGCN_1d(residual=False) and in_channels != out_channels (which drops the residual, gated_gcn_full.py:41-42,
    124-125,151-152): the oracle's layer against outputs and gradients of the reference's own layer
    (tests/golden/make_golden_layer.py), fp64 autograd on both sides (the fixture stores fp32)."""
    import os
    from helpers import GOLDEN, LAYER_VARIANTS, layer_variant_case, rel_l2
    from oracle import gatedgcn_oracle as orc
    z = np.load(os.path.join(GOLDEN, "layer_variants.npz"))
    for name, (cin, cout, bn, res) in LAYER_VARIANTS.items():
        c = layer_variant_case(name)
        residual = res and cin == cout
        assert bool(z[f"{name}/residual_used"]) == residual
        p = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in c["sd"].items()}
        h = torch.from_numpy(c["h0"]).double().requires_grad_(True)
        e = torch.from_numpy(c["e0"]).double().requires_grad_(True)
        h1, e1 = orc.layer_forward(p, None, torch.from_numpy(c["src"]).long(), torch.from_numpy(c["dst"]).long(), c["n"], h, e,
                                   batch_norm=bn, residual=residual)
        ((h1 * torch.from_numpy(c["wh"]).double()).sum() + (e1 * torch.from_numpy(c["we"]).double()).sum()).backward()
        got = {"h1": h1, "e1": e1, "gh": h.grad, "ge": e.grad, **{"g/" + k: v.grad for k, v in p.items()}}
        for k, v in got.items():
            want = z[f"{name}/{k}"]
            assert v.shape == want.shape, (name, k)
            d = np.abs(v.detach().numpy() - want).max()
            assert rel_l2(v.detach().numpy(), want) < 1e-6 or d < 1e-6 * max(1.0, np.abs(want).max()), (name, k, d)
