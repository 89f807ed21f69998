"""Host side of the internal node numbering (graph.py "Node numbering", gnm_graph_locality_order): the reference's
graphs number nodes by read id, not by genome position (pipeline.py:46-61,160-169; graph_parser.py:297-304), so the
index is built over a breadth-first renumbering when the caller's numbering is not local.  No GPU needed."""
import numpy as np
import torch

from gnnome_assembly_amd import AssemblyGraph, graph as G, synth


def _shuffled(reads, seed):
    src, dst, n = synth.make_graph(reads, seed)
    p = np.random.default_rng(seed + 5).permutation(n).astype(np.int32)     # caller id of node v: p[v]
    return src, dst, n, p


def test_locality_order_is_a_permutation_and_restores_locality():
    src, dst, n, p = _shuffled(20000, 1)
    g = AssemblyGraph(p[src], p[dst], n)
    ix = g.host_index()
    info = g.relabel_info
    assert info["mode"] == "auto" and info["relabelled"]
    assert info["local_edge_fraction"] < 0.2                     # random ids: |s - d| <= 2048 of 40000 nodes by chance only
    assert info["local_edge_fraction_after"] > 0.99              # the 0.5 % repeat edges stay long
    assert info["triangle_edge_fraction"] > 0.98
    order, rank = ix["nperm"], ix["nrank"]
    assert np.array_equal(np.sort(order), np.arange(n))
    assert np.array_equal(rank[order], np.arange(n))
    # the index is the plain index of the renumbered graph; perm still maps to the CALLER's edge ids
    s2, d2 = rank[p[src]], rank[p[dst]]
    assert np.array_equal(ix["isrc"], s2[ix["perm"]]) and np.array_equal(ix["idst"], d2[ix["perm"]])
    assert np.all(np.diff(ix["idst"]) >= 0)
    want = AssemblyGraph(s2, d2, n, node_order="keep").host_index()
    for k in ("perm", "isrc", "idst", "in_ptr", "out_ptr", "out_pos", "out_dst"):
        assert np.array_equal(ix[k], want[k]), k


def test_auto_keeps_a_numbering_that_is_already_local_and_small_graphs():
    src, dst, n = synth.make_graph(20000, 0)
    g = AssemblyGraph(src, dst, n)
    ix = g.host_index()
    assert not g.relabel_info["relabelled"] and g.relabel_info["local_edge_fraction"] > 0.99
    assert "nperm" not in ix
    s, d, m = synth.tiny_edge_case_graph(0)                        # every fixture-sized graph keeps its numbering
    t = AssemblyGraph(s, d, m)
    assert "nperm" not in t.host_index() and not t.relabel_info["relabelled"]


def test_forced_modes_and_edge_cases():
    s, d, m = synth.tiny_edge_case_graph(0)                        # self loops, duplicates, isolated nodes, hubs
    ix = AssemblyGraph(s, d, m, node_order="bfs").host_index()
    assert np.array_equal(np.sort(ix["nperm"]), np.arange(m)) and np.array_equal(ix["nrank"][ix["nperm"]], np.arange(m))
    assert np.array_equal(ix["idst"], ix["nrank"][d][ix["perm"]])
    e = AssemblyGraph(np.zeros(0, np.int32), np.zeros(0, np.int32), 5, node_order="bfs").host_index()   # no edges at all
    assert np.array_equal(e["nperm"], np.arange(5))
    src, dst, n, p = _shuffled(5000, 2)
    k = AssemblyGraph(p[src], p[dst], n, node_order="keep")
    assert "nperm" not in k.host_index() and k.relabel_info == {"mode": "keep", "relabelled": False}
    old = G.NODE_ORDER
    try:
        G.set_node_order("keep")
        assert "nperm" not in AssemblyGraph(p[src], p[dst], n).host_index()
        G.set_node_order("bfs")
        assert "nperm" in AssemblyGraph(src, dst, n).host_index()
    finally:
        G.set_node_order(old)


def test_a_path_graph_without_triangles_still_gets_ordered():
    """No triangle-supported edges at all (each read overlaps only the next): the sweep falls back to every edge."""
    n = 6000
    p = np.random.default_rng(3).permutation(n).astype(np.int32)
    src, dst = p[np.arange(n - 1)], p[np.arange(1, n)]
    g = AssemblyGraph(src, dst, n)
    ix = g.host_index()
    assert g.relabel_info["relabelled"] and g.relabel_info["triangle_edge_fraction"] == 0.0
    assert np.all(np.abs(ix["nrank"][src].astype(np.int64) - ix["nrank"][dst]) == 1)       # the path, end to end


def test_tensor_index_with_a_given_internal_numbering():
    src, dst, n, p = _shuffled(3000, 4)
    host = AssemblyGraph(p[src], p[dst], n, node_order="bfs").host_index()
    ts, td = torch.from_numpy(p[src]), torch.from_numpy(p[dst])
    idx = G.tensor_index(ts, td, n, torch.from_numpy(host["nrank"]))
    for k, v in host.items():
        assert np.array_equal(idx[k].numpy(), v), k


def test_induced_subgraph_inherits_the_parents_internal_order():
    from gnnome_assembly_amd import cluster
    src, dst, n, p = _shuffled(8000, 6)
    g = AssemblyGraph(p[src], p[dst], n)
    g.ndata["pe"] = torch.arange(n, dtype=torch.float32)[:, None]
    mask = torch.from_numpy(np.random.default_rng(0).random(n) < 0.5)
    sub = cluster.induced_subgraph(g, mask)
    ix = sub.index()
    nid = sub.ndata[cluster.NID]
    assert torch.equal(nid, torch.nonzero(mask).squeeze(1))          # the DGL surface: ascending original id
    pr = torch.from_numpy(g.host_index()["nrank"]).long()
    # internal order of the sub-graph = the parent's internal order of the kept nodes
    assert torch.all(torch.diff(pr[nid[ix["nperm"].long()]]) > 0)
    assert torch.equal(ix["nrank"].long()[ix["nperm"].long()], torch.arange(nid.numel()))
