"""ClusterGCN mini-batch mode, host logic (train.py:288-293): partitioner
properties and the exact `g.subgraph` semantics against a brute-force restatement."""
import numpy as np
import pytest
import torch

from gnnome_assembly_amd import AssemblyGraph, cluster, synth


def _graph(reads=400, seed=3, shuffle_nodes=False):
    src, dst, n = synth.make_graph(reads, seed=seed, permute_edge_ids=True)
    if shuffle_nodes:   # real read ids are not position-sorted
        relabel = np.random.default_rng(seed).permutation(n).astype(np.int32)
        src, dst = relabel[src], relabel[dst]
    return AssemblyGraph(src, dst, n), src, dst, n


@pytest.mark.parametrize("method", ["locality", "rcm", "order"])
@pytest.mark.parametrize("parts", [1, 7, 50])
def test_partition_is_balanced_total_and_deterministic(method, parts):
    g, src, dst, n = _graph()
    p1 = cluster.partition_graph(g, parts, method)
    p2 = cluster.partition_graph(AssemblyGraph(src, dst, n), parts, method)
    assert p1.shape == (n,) and p1.min() == 0 and p1.max() == parts - 1
    assert np.array_equal(p1, p2)
    sizes = np.bincount(p1, minlength=parts)
    assert sizes.max() - sizes.min() <= 1


def test_rcm_recovers_the_band_when_node_ids_are_shuffled():
    g, src, dst, n = _graph(reads=1500, shuffle_nodes=True)
    e = src.size
    cut_rcm = cluster.edge_cut(g, cluster.partition_graph(g, 20, "rcm"))
    cut_loc = cluster.edge_cut(g, cluster.partition_graph(g, 20, "locality"))
    cut_ids = cluster.edge_cut(g, cluster.partition_graph(g, 20, "order"))
    print(f"edge cut: locality {cut_loc}/{e}  rcm {cut_rcm}/{e}  node-id blocks {cut_ids}/{e}")
    assert cut_ids > 0.8 * e          # shuffled ids: almost every edge is cut
    assert cut_rcm < 0.12 * e         # the ordering finds the 1-D structure again
    assert cut_loc < 0.12 * e         # ... and so does the index's own breadth-first order (the default: host C++, linear in E)
    g2, *_ = _graph(reads=1500)
    assert cluster.edge_cut(g2, cluster.partition_graph(g2, 20, "order")) < 0.08 * e


def test_more_parts_than_nodes_and_bad_arguments():
    g = AssemblyGraph(np.array([0, 1], np.int32), np.array([1, 2], np.int32), 3)
    p = cluster.partition_graph(g, 10)
    assert sorted(p.tolist()) == [0, 1, 2]
    with pytest.raises(ValueError):
        cluster.partition_graph(g, 0)
    with pytest.raises(ValueError):
        cluster.partition_graph(g, 2, "metis")
    with pytest.raises(ValueError):
        cluster.induced_subgraph(g, torch.ones(2, dtype=torch.bool))


def test_induced_subgraph_matches_brute_force():
    src, dst, n = synth.tiny_edge_case_graph() if hasattr(synth, "tiny_edge_case_graph") else synth.make_graph(40, 0)
    g = AssemblyGraph(src, dst, n)
    e = src.size
    rng = np.random.default_rng(0)
    g.ndata["pe"] = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32))
    g.edata["y"] = torch.from_numpy(rng.integers(0, 2, e).astype(np.float32))
    mask = rng.random(n) < 0.6
    sub = cluster.induced_subgraph(g, torch.from_numpy(mask))
    nodes = np.flatnonzero(mask)
    new = {int(v): i for i, v in enumerate(nodes)}
    want = [(new[int(s)], new[int(d)], k) for k, (s, d) in enumerate(zip(src, dst)) if mask[s] and mask[d]]
    s_sub, d_sub = sub.edges()
    assert sub.num_nodes() == nodes.size and sub.num_edges() == len(want)
    assert [(int(a), int(b)) for a, b in zip(s_sub, d_sub)] == [(a, b) for a, b, _ in want]
    assert sub.edata[cluster.EID].tolist() == [k for _, _, k in want]          # ascending original edge id
    assert sub.ndata[cluster.NID].tolist() == nodes.tolist()
    assert torch.equal(sub.ndata["pe"], g.ndata["pe"][torch.from_numpy(nodes)])
    assert torch.equal(sub.edata["y"], g.edata["y"][sub.edata[cluster.EID]])


def test_loader_visits_every_cluster_once_and_keeps_only_intra_batch_edges():
    g, src, dst, n = _graph(reads=600)
    part = cluster.partition_graph(g, 23)
    gen = torch.Generator().manual_seed(5)
    loader = cluster.ClusterBatchLoader(g, part, 5, shuffle=True, generator=gen)
    assert len(loader) == 5
    seen_nodes, seen_edges, batch_of = [], [], np.full(23, -1)
    for b, sub in enumerate(loader):
        nid = sub.ndata[cluster.NID].numpy()
        cl = np.unique(part[nid])
        assert (batch_of[cl] == -1).all()
        batch_of[cl] = b
        seen_nodes.append(nid)
        seen_edges.append(sub.edata[cluster.EID].numpy())
    assert np.array_equal(np.sort(np.concatenate(seen_nodes)), np.arange(n))
    intra = batch_of[part[src]] == batch_of[part[dst]]
    assert np.array_equal(np.sort(np.concatenate(seen_edges)), np.flatnonzero(intra))
    # no shuffle: clusters in order
    first = next(iter(cluster.ClusterBatchLoader(g, part, 5, shuffle=False)))
    assert set(np.unique(part[first.ndata[cluster.NID].numpy()])) == {0, 1, 2, 3, 4}
