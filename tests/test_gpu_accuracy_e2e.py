"""End-to-end accuracy of the three matmul modes at H = 128 and H = 256 (VERDICT r5 "weak" 1 / item 5): where does a split-mode
forward leave the fp64 oracle, kernel by kernel down the stack, and how many relu decisions fall the other way?

Per (width, mode) and per layer, against the oracle evaluated in fp64 on the same graph and parameters:
  P, t, z     rel-L2 of the three saved tensors a layer's matrix kernels and sweeps produce (P: node projections; t: edge B_3
              product + gathers; z: both aggregations) -- the first row that stands out names the kernel;
  flips u/w   relu decisions of the edge / node BatchNorm outputs that differ from the fp64 run's (each one moves the gradients
              downstream by a rank-one term: what the branch-exact comparison of the parity suite factors out);
  |u| at flip the largest fp64 |u| among the flipped elements, in units of the BatchNorm output's scale: flips are legitimate
              only within rounding distance of the kink;
and for the whole model the logits' rel-L2 and the tally of gradient tensors inside the plain bar (rel-L2 <= 2e-4).  The same
columns for the oracle itself run in fp32 (torch CPU: the reference's own arithmetic) are the yardstick: a mode that flips no
more decisions than the reference's fp32 run flips is not "further from fp64", whatever the clause tally says.
Writes gpurun_out/accuracy_e2e.txt (committed copy: profiles/r06_accuracy_e2e.txt)."""
import os

import numpy as np
import pytest
import torch

from helpers import sd_to_torch, rel_l2, device_masks
from oracle import gatedgcn_oracle as orc

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(128, 3), (256, 2)]
LINES = []


def _fp64(src, dst, n, inp, sd):
    with torch.no_grad():
        s, l, g, dbg = orc.manual_forward_backward(sd_to_torch(sd, torch.float64), torch.from_numpy(src), torch.from_numpy(dst), n,
                                                   torch.from_numpy(inp["e"]).double(), torch.from_numpy(inp["pe"]).double(),
                                                   torch.from_numpy(inp["y"]).double(), float(inp["pos_weight"]), keep=True)
    return s, g, dbg


def _fp32_oracle(src, dst, n, inp, sd):
    with torch.no_grad():
        s, l, g, dbg = orc.manual_forward_backward(sd_to_torch(sd, torch.float32), torch.from_numpy(src), torch.from_numpy(dst), n,
                                                   torch.from_numpy(inp["e"]), torch.from_numpy(inp["pe"]), torch.from_numpy(inp["y"]),
                                                   float(inp["pos_weight"]), keep=True)
    return s, g, dbg


def _row(tag, L, layer_cols, logits, plain, total):
    out = [f"  {tag:18s} logits {logits:.2e}   gradient tensors within 2e-4: {plain:3d} / {total}"]
    for i in range(L):
        c = layer_cols[i]
        out.append(f"      layer {i}:  P {c['P']:.2e}  t {c['t']:.2e}  z {c['z']:.2e}   flips u {c['fu']:5d} / {c['nu']}  w {c['fw']:4d} / {c['nw']}"
                   f"   |u64| at a flip <= {c['mu']:.1e}, |w64| <= {c['mw']:.1e} (BatchNorm outputs are O(1))")
    return out


@pytest.mark.parametrize("H,L", CASES)
def test_accuracy_end_to_end_by_mode(H, L):
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import _lib, engine, synth
    dev = torch.device("cuda:0")
    src, dst, n = synth.make_graph(700, seed=H + L, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, seed=H)
    sd = synth.synth_state_dict(H, L, seed=L)
    s64, g64, d64 = _fp64(src, dst, n, inp, sd)
    u64 = [d64[i]["u"] for i in range(L)]
    w64 = [d64[i]["w"] for i in range(L)]
    E = src.size
    LINES.append(f"H = {H}, L = {L}, N = {n}, E = {E} (the graph and parameters of test_other_widths_and_norms_vs_oracle[*-{H}-{L}-True])")

    def cols_from(P_l, t_l, z_l, um, wm):
        cols = []
        for i in range(L):
            fu, fw = um[i] != (u64[i] > 0), wm[i] != (w64[i] > 0)
            cols.append(dict(P=rel_l2(P_l[i], d64[i]["P"].numpy()), t=rel_l2(t_l[i], d64[i]["t"].numpy()), z=rel_l2(z_l[i], d64[i]["z"].numpy()),
                             fu=int(fu.sum()), nu=fu.numel(), fw=int(fw.sum()), nw=fw.numel(),
                             mu=float(u64[i][fu].abs().max()) if fu.any() else 0.0, mw=float(w64[i][fw].abs().max()) if fw.any() else 0.0))
        return cols

    # the yardstick: the oracle's own arithmetic in fp32
    s32, g32, d32 = _fp32_oracle(src, dst, n, inp, sd)
    cols = cols_from([d32[i]["P"].numpy() for i in range(L)], [d32[i]["t"].numpy() for i in range(L)], [d32[i]["z"].numpy() for i in range(L)],
                     [d32[i]["u"] > 0 for i in range(L)], [d32[i]["w"] > 0 for i in range(L)])
    plain = sum(rel_l2(g32[k].numpy(), g64[k].numpy()) <= 2e-4 for k in g64)
    LINES.extend(_row("oracle in fp32", L, cols, rel_l2(s32.numpy(), s64.numpy()), plain, len(g64)))
    flips32 = sum(c["fu"] + c["fw"] for c in cols)

    worst = {}
    for mode in ("f32", "bf16x3", "f16x2"):
        _lib.set_matmul_mode(mode)
        try:
            g = G.AssemblyGraph(src, dst, n).to(dev)
            P = {k: v.to(dev) for k, v in sd_to_torch(sd).items()}
            scores, ms = engine.model_forward(g, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev), P, L, True)
            idx = g.index()
            perm = idx["perm"].long().cpu()
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())
            nrank = idx["nrank"].long().cpu() if "nrank" in idx else None
            nodes = (lambda x: x.cpu() if nrank is None else x.cpu()[nrank])
            P_l = [nodes(s.P).numpy() for s in ms.layers]
            t_l = [s.t.cpu()[inv].numpy() for s in ms.layers]
            z_l = [nodes(s.z).numpy() for s in ms.layers]
            masks = device_masks(ms, sd, inp["e"], idx)
            loss, gs = engine.bce_with_logits(scores, torch.from_numpy(inp["y"]).to(dev), float(inp["pos_weight"]))
            Gd = engine.model_backward(g, P, L, ms, gs)
            torch.cuda.synchronize()
        finally:
            _lib.set_matmul_mode(_lib.DEFAULT_MATMUL_MODE)
        cols = cols_from(P_l, t_l, z_l, masks["u"], masks["w"])
        plain = sum(rel_l2(Gd[k].cpu().numpy(), g64[k].numpy()) <= 2e-4 for k in g64)
        lg = rel_l2(scores.cpu().numpy(), s64.numpy())
        LINES.extend(_row(mode, L, cols, lg, plain, len(g64)))
        worst[mode] = (lg, max(max(c["P"], c["t"], c["z"]) for c in cols), sum(c["fu"] + c["fw"] for c in cols),
                       max(max(c["mu"], c["mw"]) for c in cols))
    LINES.append("")
    for mode, (lg, fwd, flips, kink) in worst.items():
        # every forward tensor at fp32 round-off, logits two decades inside the 1e-4 bar, and every flipped decision within rounding
        # distance of the kink (a flip far from zero would be an error, not a rounding)
        assert lg <= 5e-6 and fwd <= 5e-6, (mode, lg, fwd)
        assert kink <= 1e-4, (mode, kink)
        # no mode flips an order of magnitude more decisions than the reference's own fp32 arithmetic does
        assert flips <= 10 * max(flips32, 5), (mode, flips, flips32)


def test_zz_write_accuracy_table():
    if not LINES:
        pytest.skip("no accuracy case ran in this session")
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    head = ["# end-to-end accuracy by matmul mode (tests/test_gpu_accuracy_e2e.py); every number is a distance from the fp64 oracle",
            "# rel-L2 of the saved forward tensors per layer, relu decisions that differ from the fp64 run, gradient tensors inside the plain bar", ""]
    open(os.path.join(out, "accuracy_e2e.txt"), "w").write("\n".join(head + LINES) + "\n")
