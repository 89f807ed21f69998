"""CPU: the C-ABI library loads and exports every symbol include/gnm.h declares; host-side
logic (graph index, module surface, state_dict schema, fail-loud behaviour)."""
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build(verbose=False)
    from gnnome_assembly_amd import _lib
    return _lib.load()


def test_every_header_symbol_is_exported_and_bound(lib):
    from gnnome_assembly_amd import _lib
    hdr = open(os.path.join(REPO, "include", "gnm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gnm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gnm_abi_version() == _lib.ABI_VERSION == 7
    assert lib.gnm_max_partial_blocks() == 2048


def test_invalid_arguments_return_error_not_crash(lib):
    import ctypes as C
    from gnnome_assembly_amd import _lib
    rc = lib.gnm_graph_build_index(None, None, 4, 3, None, None, None, None, None, None, None)
    assert rc < 0 and b"null" in lib.gnm_last_error()
    src = np.array([0, 9], np.int32)
    dst = np.array([1, 1], np.int32)
    out = [np.zeros(8, np.int32) for _ in range(7)]
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.gnm_graph_build_index(p(src), p(dst), 4, 2, *[p(a) for a in out])
    assert rc == -2 and b"outside" in lib.gnm_last_error()
    with pytest.raises(_lib.GnmError):
        _lib.check(rc, "gnm_graph_build_index")
    assert lib.gnm_gemm_f32(7, 1, 1, 1, None, 1, None, 1, None, 1, None, None, 0, 0, None, 0, None) < 0


@pytest.mark.parametrize("kind", ["tiny", "banded", "banded_perm", "empty"])
def test_graph_index(lib, kind):
    from gnnome_assembly_amd import AssemblyGraph, synth
    if kind == "tiny":
        s, d, n = synth.tiny_edge_case_graph(1)
    elif kind == "empty":
        s, d, n = np.zeros(0, np.int32), np.zeros(0, np.int32), 5
    else:
        s, d, n = synth.make_graph(3000, 2, permute_edge_ids=(kind == "banded_perm"))
    g = AssemblyGraph(s, d, n)
    ix = g.host_index()
    perm = ix["perm"]
    assert np.array_equal(np.argsort(d, kind="stable").astype(np.int32), perm)
    assert np.array_equal(s[perm], ix["isrc"]) and np.array_equal(d[perm], ix["idst"])
    assert np.array_equal(np.diff(ix["in_ptr"]), np.bincount(d, minlength=n))
    assert np.array_equal(np.diff(ix["out_ptr"]), np.bincount(s, minlength=n))
    assert np.array_equal(ix["isrc"][ix["out_pos"]], np.repeat(np.arange(n), np.diff(ix["out_ptr"])))
    assert np.array_equal(ix["idst"][ix["out_pos"]], ix["out_dst"])
    for v in range(min(n, 50)):   # ascending internal position inside a source
        seg = ix["out_pos"][ix["out_ptr"][v]:ix["out_ptr"][v + 1]]
        assert np.all(np.diff(seg) > 0)
    assert g.num_nodes() == n and g.num_edges() == s.size
    es, ed = g.edges()
    assert np.array_equal(es.numpy(), s) and np.array_equal(ed.numpy(), d)
    assert np.array_equal(g.in_degrees().numpy(), np.bincount(d, minlength=n))


def test_module_surface_and_state_dict_schema():
    """Constructor signatures and state_dict keys/shapes of the reference
    (full_graph.py:12-20) so that reference checkpoints load."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    for H, L, nparam in ((64, 1, 39985), (128, 8, 826033)):
        m = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
        sd = m.state_dict()
        ref = synth.synth_state_dict(H, L, 0)
        assert list(sd.keys()) == list(ref.keys())
        assert all(tuple(sd[k].shape) == ref[k].shape for k in ref)
        assert sum(p.numel() for p in m.parameters()) == nparam
        assert not any("running" in k or "num_batches" in k for k in sd)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    assert G.layers.GraphGatedGCN(2, 32, True).convs[1].A_1.weight.shape == (32, 32)
    assert G.layers.ScorePredictor(32, 64).W1.weight.shape == (64, 96)
    G.layers.NodeEncoder(1, 8), G.layers.EdgeEncoder(2, 8)
    assert G.layers.GatedGCN_1d(32, 32, False).batch_norm is False     # LayerNorm mode exists
    wide = G.layers.GatedGCN_1d(32, 64, True)                         # in != out: the residual is dropped (gated_gcn_full.py:41-42)
    assert wide.residual is False and wide.B_3.weight.shape == (64, 32)
    assert G.layers.GatedGCN_1d(32, 32, True, residual=False).residual is False
    assert G.layers.GatedGCN_1d(32, 48, True).B_3.weight.shape == (48, 32)     # any width (run zero-padded to the next of layers.RUN_WIDTHS)
    assert G.layers.padded_width(96) == 128 and G.layers.padded_width(128) == 128
    assert G.layers.GatedGCN_1d(32, 300, True).B_3.weight.shape == (300, 32)   # wider than the widest kernel: 256-column chunks
    with pytest.raises(NotImplementedError):
        G.layers.GatedGCN_1d(32, 300, False)                           # ... BatchNorm only (LayerNorm rows span the chunks)


def test_product_path_has_no_cpu_fallback():
    """CPU tensors must raise: the HIP path is the only path."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth, _lib
    s, d, n = synth.tiny_edge_case_graph(0)
    inp = synth.make_inputs(s, d, n)
    m = G.GraphGatedGCNModel(1, 2, 32, 16, 1, 64, True, 16)
    g = G.AssemblyGraph(s, d, n)
    with pytest.raises(_lib.GnmError):
        m(g, None, torch.from_numpy(inp["e"]), torch.from_numpy(inp["pe"]))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(REPO, "gnnome_assembly_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("CPU oracle", ""), f


def test_harness_metrics_match_reference_definitions():
    """tfpn_counts == utils.calculate_tfpn, calculate_metrics keeps the reference's (swapped) naming
    (utils.py:217-240); golden TP/TN/FP/FN of the reference run are in the fixtures."""
    import numpy as np
    from gnnome_assembly_amd import train as T
    z = np.load(os.path.join(REPO, "tests", "golden", "tiny_h64l1_s0.npz"))
    pred = torch.from_numpy(z["scores64"]).reshape(-1)
    y = torch.from_numpy(z["y"]).double()
    assert T.tfpn_counts(pred, y).tolist() == [int(v) for v in z["tfpn"]]
    TP, TN, FP, FN = (int(v) for v in z["tfpn"])
    assert np.allclose(T.calculate_metrics(TP, TN, FP, FN), z["metrics"])
    assert T.calculate_metrics(0, 5, 0, 0) == (1.0, 0, 0, 0)
    assert abs(T.get_hyperparameters()["decay"] - 0.95) < 1e-12


def test_tensor_index_equals_host_index():
    """graph.tensor_index (tensor ops, used for graphs born on a device) == gnm_graph_build_index (host)."""
    import numpy as np
    import torch
    from gnnome_assembly_amd import AssemblyGraph, synth
    from gnnome_assembly_amd.graph import tensor_index
    cases = [synth.make_graph(300, 1, permute_edge_ids=True), synth.tiny_edge_case_graph()]
    cases.append((np.array([2, 2, 0], np.int32), np.array([2, 0, 1], np.int32), 5))     # isolated nodes, self loop
    for src, dst, n in cases:
        want = AssemblyGraph(src, dst, n).host_index()
        got = tensor_index(torch.from_numpy(np.asarray(src, np.int32)), torch.from_numpy(np.asarray(dst, np.int32)), int(n))
        for k, v in want.items():
            assert np.array_equal(got[k].numpy(), v), k
        g = AssemblyGraph.from_tensors(torch.from_numpy(np.asarray(src, np.int32)), torch.from_numpy(np.asarray(dst, np.int32)), int(n))
        assert g.num_edges() == len(src) and g.num_nodes() == n
        for k, v in want.items():
            assert np.array_equal(g.index()[k].numpy(), v), k
        assert np.array_equal(g._src, np.asarray(src, np.int32))        # lazy host copy
        # a GIVEN internal numbering (mini-batch sub-graphs inherit the parent's): the host index follows it as the
        # device-side index does, whichever of the two is built first
        rank = torch.from_numpy(np.random.default_rng(3).permutation(int(n)).astype(np.int32))
        s_t, d_t = torch.from_numpy(np.asarray(src, np.int32)), torch.from_numpy(np.asarray(dst, np.int32))
        dev_first = AssemblyGraph.from_tensors(s_t, d_t, int(n), nrank=rank).index()
        host_first = AssemblyGraph.from_tensors(s_t, d_t, int(n), nrank=rank)
        hidx = host_first.host_index()
        assert set(hidx) == set(dev_first)
        for k, v in hidx.items():
            assert np.array_equal(dev_first[k].numpy(), v), k
        assert all(np.array_equal(host_first.index()[k].numpy(), v) for k, v in hidx.items())


def test_graph_file_round_trip(tmp_path):
    import numpy as np
    import torch
    from gnnome_assembly_amd import AssemblyGraph, io, synth
    src, dst, n = synth.make_graph(200, 2, permute_edge_ids=True)
    g = AssemblyGraph(src, dst, n)
    rng = np.random.default_rng(0)
    g.ndata["read_length"] = torch.from_numpy(rng.integers(1, 9, n))
    g.ndata["pe"] = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32))
    g.edata["score"] = torch.from_numpy(rng.standard_normal(src.size).astype(np.float32))
    path = str(tmp_path / "g.npz")
    io.save_graph(path, g)
    h = io.load_graph(path)
    assert h.num_nodes() == n and np.array_equal(h._src, src) and np.array_equal(h._dst, dst)
    assert set(h.ndata) == {"read_length", "pe"} and set(h.edata) == {"score"}
    for k in g.ndata:
        assert torch.equal(h.ndata[k], g.ndata[k])
    assert torch.equal(h.edata["score"], g.edata["score"])
    bad = dict(np.load(path))
    bad["edata/score"] = bad["edata/score"][:-1]
    np.savez(str(tmp_path / "bad.npz"), **bad)
    import pytest
    with pytest.raises(ValueError):
        io.load_graph(str(tmp_path / "bad.npz"))


def test_from_dgl_and_foreign_graph_wrapping():
    """from_dgl / as_assembly_graph against the committed DGL stand-in (tests/golden/dgl_standin: the graph type the golden
    vectors were generated with): structure in edge-id order, ndata / edata carried over, the wrapper is built once and
    cached on the foreign object (train.py:252 passes the DGLGraph itself to the model)."""
    import sys
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    from gnnome_assembly_amd.graph import as_assembly_graph
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dgl_standin"))
    try:
        import dgl
    finally:
        sys.path.pop(0)
    s, d, n = synth.tiny_edge_case_graph(1)
    dg = dgl.DGLGraph(s, d, n)
    dg.ndata["x"] = torch.ones(n, 1)
    dg.edata["y"] = torch.arange(s.size, dtype=torch.float32)
    ag = G.from_dgl(dg)
    ss, dd = ag.edges()
    assert ag.num_nodes() == n and ag.num_edges() == s.size
    assert np.array_equal(ss.numpy(), s) and np.array_equal(dd.numpy(), d)
    assert torch.equal(ag.ndata["x"], dg.ndata["x"]) and torch.equal(ag.edata["y"], dg.edata["y"])
    ref = G.AssemblyGraph(s, d, n).host_index()
    got = ag.host_index()
    assert all(np.array_equal(ref[k], got[k]) for k in ref)
    w1 = as_assembly_graph(dg)
    assert as_assembly_graph(dg) is w1 and as_assembly_graph(w1) is w1
    with pytest.raises(TypeError):
        as_assembly_graph(object())
    # the reference's loops do g = g.to(device) every step (train.py:244,297): DGL hands back a NEW object each time -- the
    # STRUCTURE (host index, locality order, sweep plans) must be found again by the content of the edge list, not rebuilt;
    # the features are the caller's own: two live graphs of equal structure never see each other's ndata / edata
    from gnnome_assembly_amd import graph as gmod
    old = gmod.GRAPH_CACHE_MIN_EDGES
    gmod.GRAPH_CACHE_MIN_EDGES = 0            # the tiny test graph would otherwise be below the caching threshold
    gmod._BY_CONTENT.clear()
    try:
        dgA = dgl.DGLGraph(s.copy(), d.copy(), n)
        dgA.edata["y"] = torch.full((s.size,), 7.0)
        wA = as_assembly_graph(dgA)
        dg2 = dgl.DGLGraph(s.copy(), d.copy(), n)
        dg2.edata["y"] = torch.zeros(s.size)
        w2 = as_assembly_graph(dg2)
        assert w2 is not wA and w2._dev_index is wA._dev_index and w2._plans is wA._plans, "same edge list: shared structure"
        assert w2.host_index() is wA.host_index()
        assert torch.equal(w2.edata["y"], dg2.edata["y"]) and torch.equal(wA.edata["y"], dgA.edata["y"]), \
            "each wrapper carries its own caller's features"
        assert len(gmod._BY_CONTENT) == 1 and not next(iter(gmod._BY_CONTENT.values()))[0].edata, "the cache keeps no features"
        p = np.random.default_rng(0).permutation(s.size)
        w3 = as_assembly_graph(dgl.DGLGraph(s[p], d[p], n))
        assert w3._dev_index is not wA._dev_index, "another edge-id order is another graph"
        # a fingerprint collision must not hand a different graph a stale index: force one
        real = gmod._fingerprint
        gmod._fingerprint = lambda g: (0, 0, 0, 0)
        try:
            gmod._BY_CONTENT.clear()
            c1 = as_assembly_graph(dgl.DGLGraph(s.copy(), d.copy(), n))
            c2 = as_assembly_graph(dgl.DGLGraph(s[p], d[p], n))
            assert c2._dev_index is not c1._dev_index
            assert np.array_equal(c2.edges()[0].numpy(), s[p]) and np.array_equal(c2.host_index()["perm"],
                                                                                   G.AssemblyGraph(s[p], d[p], n).host_index()["perm"])
        finally:
            gmod._fingerprint = real
        # bounded by bytes as well as by entries
        gmod._BY_CONTENT.clear()
        oldb = gmod.GRAPH_CACHE_BYTES
        gmod.GRAPH_CACHE_BYTES = 1
        try:
            as_assembly_graph(dgl.DGLGraph(s.copy(), d.copy(), n))
            as_assembly_graph(dgl.DGLGraph(s[p], d[p], n))
            assert len(gmod._BY_CONTENT) == 1
        finally:
            gmod.GRAPH_CACHE_BYTES = oldb
        # an object that takes no attributes and no weak references is wrapped but never cached by id (ids are reused)

        class Frozen:
            __slots__ = ("s", "d", "n")

            def __init__(self, s_, d_, n_):
                self.s, self.d, self.n = s_, d_, n_

            def edges(self):
                return torch.from_numpy(self.s), torch.from_numpy(self.d)

            def num_nodes(self):
                return self.n

            def num_edges(self):
                return self.s.size
        gmod._BY_CONTENT.clear()
        w0 = as_assembly_graph(dgl.DGLGraph(s.copy(), d.copy(), n))
        fz = Frozen(s, d, n)
        assert as_assembly_graph(fz)._dev_index is w0._dev_index and id(fz) not in gmod._WRAPPED
    finally:
        gmod.GRAPH_CACHE_MIN_EDGES = old
        gmod._BY_CONTENT.clear()
    # below the threshold a foreign graph is neither fingerprinted nor kept (rebuilding is cheaper than the synchronisation)
    as_assembly_graph(dgl.DGLGraph(s.copy(), d.copy(), n))
    assert len(gmod._BY_CONTENT) == 0


def test_no_undefined_names_in_the_package():
    """The HIP-only paths of the package never run in the build container: a name that is read without being bound anywhere
    (a NameError on the GPU box) is caught here instead (tools/lint_names.py)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "lint_names.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_schedule_switches_are_an_options_object_scoped_to_the_thread_and_kept_with_the_saved_state():
    """engine.Options ("a small options object ... instead of module globals"): immutable, `options(...)` / `use(opts)`
    change what is current for the CALLING thread only, a pass takes `opts=`, and a backward pass without one runs under the options
    its forward left in the saved state -- autograd runs backward on another thread, where a `with options(...)` of the caller
    would not be seen."""
    import threading
    from gnnome_assembly_amd import _lib, engine
    base = engine.current()
    assert isinstance(base, engine.Options) and engine.CHAIN == base.CHAIN and engine.ACTIVATIONS == base.ACTIVATIONS
    with pytest.raises(AttributeError):
        base.CHAIN = False
    with pytest.raises(_lib.GnmError):
        base.replace(NO_SUCH_SWITCH=1)
    with pytest.raises(_lib.GnmError):
        base.replace(ACTIVATIONS="half")
    with pytest.raises(_lib.GnmError):
        base.replace(TN_AT="later")
    # TN_AT = "auto" (the default): where the deferred weight-gradient launch goes is decided by graph size (engine.tn_at)
    with engine.options(TN_AT="auto"):
        assert engine.tn_at(engine.TN_AT_NOW_NODES) == "now" and engine.tn_at(engine.TN_AT_NOW_NODES - 1) == "next"
        assert engine.tn_at(1_500_000) == "now" and engine.tn_at(220_000) == "next"       # the metric's graph / the true chr19 size
    with engine.options(TN_AT="next"):
        assert engine.tn_at(1_500_000) == "next"
    with engine.options(TN_AT="now"):
        assert engine.tn_at(1000) == "now"
    seen = []
    with engine.options(CHAIN=not base.CHAIN, TN_AT="next") as o:
        assert engine.current() is o and engine.CHAIN == (not base.CHAIN) and o.TN_AT == "next"
        t = threading.Thread(target=lambda: seen.append(engine.current()))
        t.start()
        t.join()
        with engine.options(TWO_SIDED=False) as o2:                       # nested: on top of the enclosing block
            assert o2.CHAIN == (not base.CHAIN) and o2.TWO_SIDED is False
        assert engine.current() is o
    assert seen == [base] and engine.current() is base                    # the other thread never saw the block; restored on exit

    class Saved:
        opts = base.replace(NODE_FUSED=not base.NODE_FUSED)

    @engine._scoped(1)
    def fake_backward(x, saved):
        return engine.current()
    assert fake_backward(0, Saved()) is Saved.opts                        # the forward's options, from the saved state
    assert fake_backward(0, Saved(), opts=base) is base                   # an explicit opts= wins
    assert fake_backward(0, object()) is base
    try:
        engine.set_activation_mode("lean")
        assert engine.ACTIVATIONS == "lean" and engine.current().ACTIVATIONS == "lean"
    finally:
        engine.set_activation_mode(base.ACTIVATIONS)
    assert engine.current().ACTIVATIONS == base.ACTIVATIONS


def test_widths_above_the_widest_kernel_are_legal_for_batchnorm_layers():
    """gated_gcn_full.py:44-50: nn.Linear(in, out) takes any width.  BatchNorm layers above 256 run as 256-column problems
    (engine.WIDE_CHUNK; parity on the GPU tier: test_other_widths_and_norms_vs_oracle[320-2-True], [512-1-True]); LayerNorm rows
    span the chunks and are refused at construction with a message that says so."""
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import layers
    assert [layers.padded_width(w) for w in (1, 32, 33, 64, 96, 128, 200, 256, 257, 320, 512, 513)] == \
        [32, 32, 128, 128, 128, 128, 256, 256, 512, 512, 512, 768]      # 64 runs padded to 128 (layers.RUN_WIDTHS: measured faster)
    m = G.GraphGatedGCNModel(1, 2, 320, 16, 2, 64, True, 16)
    assert m.gnn.convs[0].A_1.weight.shape == (320, 320) and m.predictor.W1.weight.shape == (64, 960)
    with pytest.raises(NotImplementedError, match="LayerNorm"):
        G.GraphGatedGCNModel(1, 2, 512, 16, 1, 64, False, 16)
    G.layers.GatedGCN_1d(48, 256, False)          # LayerNorm up to the widest kernel width stays legal


def test_a_backward_in_another_matmul_mode_than_its_forward_is_refused():
    """the matmul mode is process-wide, not an Option; a forward records it with its activations and the backward checks it
    (a lean-mode backward rebuilds P and t with the forward's kernels; the chained schedule exists in the split modes only)."""
    from gnnome_assembly_amd import _lib, engine
    s = engine.LayerSaved(matmul="f32" if _lib.get_matmul_mode() != "f32" else "f16x2")
    with pytest.raises(_lib.GnmError, match="matmul mode"):
        engine._same_matmul_mode(s)
    engine._same_matmul_mode(engine.LayerSaved(matmul=_lib.get_matmul_mode()))
    engine._same_matmul_mode(engine.LayerSaved())          # no record (a state built by hand): unchecked
