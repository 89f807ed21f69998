"""The engine's results do not depend on how the caller numbers the nodes (graph.py "Node numbering"): the reference's
graphs come in read-id order (pipeline.py:46-61,160-169, graph_parser.py:297-304), the kernels want genome order, the
index renumbers internally and nothing node-shaped leaves in the internal numbering.  -m gpu."""
import numpy as np
import pytest
import torch

from helpers import load_case, sd_to_torch, rel_l2, assert_parity

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def test_model_is_node_relabelling_equivariant():
    """Same graph, same edge ids, three node numberings (position-sorted kept as is; position-sorted but renumbered
    breadth-first; randomly shuffled -> renumbered by 'auto'): logits, loss and gradients agree up to summation
    order."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    dev = _dev()
    H, L, seed = 128, 4, 3
    src, dst, n = synth.make_graph(30000, seed)
    inp = synth.make_inputs(src, dst, n, seed)
    sd = synth.synth_state_dict(H, L, seed)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    p = np.random.default_rng(17).permutation(n).astype(np.int32)          # caller id of node v: p[v]
    pe = inp["pe"]
    pe_shuf = np.empty_like(pe)
    pe_shuf[p] = pe
    e = torch.from_numpy(inp["e"]).to(dev)
    y = torch.from_numpy(inp["y"]).to(dev)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    graphs = {"sorted/keep": (G.AssemblyGraph(src, dst, n, node_order="keep"), pe),
              "sorted/bfs": (G.AssemblyGraph(src, dst, n, node_order="bfs"), pe),
              "shuffled/auto": (G.AssemblyGraph(p[src], p[dst], n), pe_shuf),
              "shuffled/keep": (G.AssemblyGraph(p[src], p[dst], n, node_order="keep"), pe_shuf)}
    out = {}
    for name, (g, pe_np) in graphs.items():
        g = g.to(dev)
        model.zero_grad(set_to_none=True)
        s = model(g, None, e, torch.from_numpy(pe_np).to(dev))
        loss = crit(s.squeeze(-1), y)
        loss.backward()
        out[name] = (s.detach().cpu().numpy(), loss.item(), {k: v.grad.cpu().numpy() for k, v in model.named_parameters()})
        assert ("nperm" in g.index()) == (name in ("sorted/bfs", "shuffled/auto")), name
    ref = out["sorted/keep"]
    for name in ("sorted/bfs", "shuffled/auto", "shuffled/keep"):
        s, l, g = out[name]
        assert_parity(s, ref[0], f"{name} logits", rtol=1e-4, atol=1e-5, l2=2e-5)
        assert abs(l - ref[1]) < 2e-6, (name, l, ref[1])
        for k in g:
            r = rel_l2(g[k], ref[2][k])
            assert r < 1e-3 or float(np.abs(g[k] - ref[2][k]).max()) < 2e-7, (name, k, r)


def test_standalone_modules_under_an_internal_numbering_match_the_oracle():
    """GatedGCN_1d / ScorePredictor take and return node tensors in the CALLER's numbering whatever the index uses."""
    import gnnome_assembly_amd as G
    from oracle import gatedgcn_oracle as orc
    dev = _dev()
    z, sd, H, L, bn = load_case("small_h64l1_s1.npz")
    src, dst, n = z["src"], z["dst"], int(z["n"])
    E = src.size
    rng = np.random.default_rng(12)
    h0 = rng.standard_normal((n, H)).astype(np.float32)
    e0 = rng.standard_normal((E, H)).astype(np.float32)
    graph = G.AssemblyGraph(src, dst, n, node_order="bfs").to(dev)
    assert "nperm" in graph.index()
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
    wh = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32)).to(dev)
    ((s * w).sum() + (h1 * wh).sum()).backward()
    p = sd_to_torch(sd, torch.float64, requires_grad=True)
    hh = torch.from_numpy(h0).double().requires_grad_(True)
    ee = torch.from_numpy(e0).double().requires_grad_(True)
    ts, td = torch.from_numpy(src).long(), torch.from_numpy(dst).long()
    rh, re = orc.layer_forward(p, 0, ts, td, n, hh, ee)
    rs = orc.predictor_forward(p, ts, td, rh, re)
    ((rs * w.cpu().double()).sum() + (rh * wh.cpu().double()).sum()).backward()
    for name, got, want in (("h1", h1, rh), ("e1", e1, re), ("scores", s, rs), ("d/dh", h.grad, hh.grad),
                            ("d/de", e.grad, ee.grad)):
        r = rel_l2(got.detach().cpu().numpy(), want.detach().numpy())
        assert r < 2e-4, (name, r)


def test_feature_preparation_under_an_internal_numbering():
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import features, synth
    dev = _dev()
    src, dst, n = synth.make_graph(20000, 2)
    p = np.random.default_rng(8).permutation(n).astype(np.int32)
    a = features.positional_encoding(G.AssemblyGraph(src, dst, n, node_order="keep").to(dev)).cpu().numpy()
    g = G.AssemblyGraph(p[src], p[dst], n).to(dev)
    b = features.positional_encoding(g).cpu().numpy()
    assert g.relabel_info["relabelled"]
    assert np.array_equal(b[p][:, :2], a[:, :2])                       # degrees: exact
    assert_parity(b[p][:, 2:], a[:, 2:], "pagerank pe", rtol=1e-5, atol=1e-12, l2=1e-6)
