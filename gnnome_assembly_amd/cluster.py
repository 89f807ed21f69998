"""ClusterGCN-style mini-batch mode: counterpart of `dgl.dataloading.ClusterGCNSampler` +
`DataLoader(g, arange(num_clusters), sampler, batch_size, shuffle=True)` as train.py:288-293 and
:438-439 use them.  It is the reference's DEFAULT training mode
(hyperparameters.py:15-18: 500 clusters, 50 per batch) and the escape hatch when a graph's saved
activations exceed HBM.

What DGL does there, restated: METIS splits the nodes into `num_parts` balanced clusters with a
small edge cut; a mini-batch is the union of `batch_size` clusters; the sampler returns
`g.subgraph(nodes)` -- the INDUCED subgraph with relabelled nodes, edges kept in ascending original
edge id, all `ndata` / `edata` rows carried along (so `in_deg`, `out_deg`, `pe` are the FULL graph's
values, train.py:301-305).

METIS lives inside DGL and is not available here, so the PARTITIONER is this module's own (parity
with METIS's particular cut is unpinned and not attempted): overlap graphs of a chromosome are
nearly one-dimensional, so a breadth-first (Cuthill-McKee) ordering of the symmetrised graph -- the one
the graph index numbers its nodes by, gnm_graph_locality_order -- cut
into equal contiguous blocks gives balanced parts whose edge cut is the band width per boundary.
Repeat-induced long-range edges would fold that ordering; they are recognised by having no common
neighbour and left out of the ordering (not out of the graph).
The SUBGRAPH semantics (induced edges, relabelling, edge order, feature slicing) are exact and
tested against a brute-force restatement."""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graph import AssemblyGraph

__all__ = ["node_order", "partition_graph", "edge_cut", "induced_subgraph", "ClusterBatchLoader", "NID", "EID"]

NID = "_ID"     # dgl.NID / dgl.EID: original ids of a subgraph's nodes / edges
EID = "_ID"


def node_order(graph: AssemblyGraph, method: str = "locality") -> np.ndarray:
    """The 1-D node ordering the parts are cut from (cached on the graph: only the number of parts
    changes between epochs, train.py:291)."""
    cache = graph.__dict__.setdefault("_node_order", {})
    if method not in cache:
        n = graph.num_nodes()
        if method == "order":
            cache[method] = np.arange(n, dtype=np.int64)
        elif method == "locality":
            # the same order the graph index numbers its nodes by (gnm_graph_locality_order, host C++, linear in E:
            # breadth-first over the triangle-supported overlaps from a pseudo-peripheral node of every component)
            idx = graph.host_index() if graph._host_index is not None or graph._src_t is None else None
            if idx is not None and "nperm" in idx:
                cache[method] = idx["nperm"].astype(np.int64)           # the index has it already
            else:
                import ctypes as C
                from . import _lib
                lib = _lib.load()
                src, dst = np.ascontiguousarray(graph._src, np.int32), np.ascontiguousarray(graph._dst, np.int32)
                order, rank = np.empty(n, np.int32), np.empty(n, np.int32)
                ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
                _lib.check(lib.gnm_graph_locality_order(ptr(src), ptr(dst), n, src.size, ptr(order), ptr(rank), None),
                           "gnm_graph_locality_order")
                cache[method] = order.astype(np.int64)
        elif method == "rcm":
            import scipy.sparse as sp
            from scipy.sparse.csgraph import reverse_cuthill_mckee
            src, dst = graph._src.astype(np.int64), graph._dst.astype(np.int64)
            a = sp.csr_matrix((np.ones(2 * src.size, np.int32), (np.concatenate((src, dst)), np.concatenate((dst, src)))),
                              shape=(n, n))
            a.data[:] = 1                      # duplicates were summed
            a.setdiag(0)
            a.eliminate_zeros()
            # True overlaps are transitive (reads that overlap share neighbours); repeat-induced edges
            # join reads with disjoint neighbourhoods and act as shortcuts that fold a breadth-first
            # ordering.  Order on the triangle-supported edges only (common-neighbour count > 0).
            tri = (a @ a).multiply(a)
            tri.eliminate_zeros()
            core = tri if tri.nnz >= a.nnz // 2 else a
            core.data[:] = 1
            cache[method] = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(core), symmetric_mode=True), dtype=np.int64)
        else:
            raise ValueError(f"unknown partition method {method!r}")
    return cache[method]


def partition_graph(graph: AssemblyGraph, num_parts: int, method: str = "locality") -> np.ndarray:
    """part[v] in [0, num_parts) for every node; parts are balanced to within one node.
    method 'locality': contiguous blocks of the index's own breadth-first order (default; host C++, linear in E);
           'rcm': contiguous blocks of scipy's reverse Cuthill-McKee ordering (the round-1 partitioner: E-sized sparse
                  products, kept for comparison);
           'order': contiguous blocks of the node ids themselves (reads already position-sorted)."""
    n = graph.num_nodes()
    if num_parts < 1:
        raise ValueError("num_parts must be >= 1")
    num_parts = min(num_parts, max(n, 1))
    order = node_order(graph, method)
    part = np.empty(n, np.int32)
    # block b gets positions [b*n/P, (b+1)*n/P): sizes differ by at most one
    part[order] = (np.arange(n, dtype=np.int64) * num_parts // max(n, 1)).astype(np.int32)
    return part


def edge_cut(graph: AssemblyGraph, part: np.ndarray) -> int:
    """number of edges whose end points lie in different parts"""
    return int(np.count_nonzero(part[graph._src] != part[graph._dst]))


def induced_subgraph(graph: AssemblyGraph, node_mask: torch.Tensor) -> AssemblyGraph:
    """`g.subgraph(nodes)` of DGL: nodes relabelled in ascending original id, induced edges in
    ascending original edge id, ndata / edata rows sliced, original ids in ndata[NID] / edata[EID].
    Runs on the graph's device (mask, compaction and relabelling are E- and N-sized tensor ops)."""
    dev = graph.device
    src, dst = graph.edges()
    node_mask = node_mask.to(dev)
    if node_mask.dtype != torch.bool or node_mask.numel() != graph.num_nodes():
        raise ValueError("node_mask must be a bool tensor with one entry per node")
    nid = torch.nonzero(node_mask, as_tuple=False).squeeze(1)
    new_id = torch.cumsum(node_mask.to(torch.int32), 0, dtype=torch.int32) - 1
    keep = node_mask[src.long()] & node_mask[dst.long()]
    eid = torch.nonzero(keep, as_tuple=False).squeeze(1)
    s_sub = new_id[src[eid].long()]
    d_sub = new_id[dst[eid].long()]
    # the parent's internal node order (graph.py "Node numbering"), restricted to the sub-graph: ranks of the kept nodes
    # compressed to 0 .. n_sub-1 -- so a mini-batch of a graph with scattered node ids is as local as the parent
    nrank = None
    pr = graph.index(dev).get("nrank")      # (builds the parent's index once; cached on the graph)
    if pr is not None:
        nrank = torch.empty(nid.numel(), dtype=torch.int64, device=dev)
        nrank[torch.argsort(pr[nid].long())] = torch.arange(nid.numel(), device=dev)
    sub = AssemblyGraph.from_tensors(s_sub, d_sub, int(nid.numel()), nrank)  # edges and index stay on the device
    sub.ndata = {k: v[nid] for k, v in graph.ndata.items()}
    sub.edata = {k: v[eid] for k, v in graph.edata.items()}
    sub.ndata[NID] = nid
    sub.edata[EID] = eid
    return sub


class ClusterBatchLoader:
    """Iterates the mini-batches of one graph: `batch_size` clusters per batch, clusters shuffled when
    `shuffle` (DataLoader(..., shuffle=True, drop_last=False)); yields induced subgraphs."""

    def __init__(self, graph: AssemblyGraph, part: np.ndarray, batch_size: int, shuffle: bool = True,
                 generator: Optional[torch.Generator] = None):
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        self.graph = graph
        self.part = torch.from_numpy(np.ascontiguousarray(part, dtype=np.int32)).to(graph.device)
        self.num_parts = int(part.max()) + 1 if part.size else 0
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.generator = generator

    def __len__(self) -> int:
        return (self.num_parts + self.batch_size - 1) // self.batch_size

    def batches(self) -> List[torch.Tensor]:
        ids = torch.randperm(self.num_parts, generator=self.generator) if self.shuffle else torch.arange(self.num_parts)
        return [ids[i:i + self.batch_size] for i in range(0, self.num_parts, self.batch_size)]

    def __iter__(self) -> Iterator[AssemblyGraph]:
        dev = self.graph.device
        for ids in self.batches():
            sel = torch.zeros(self.num_parts, dtype=torch.bool, device=dev)
            sel[ids.to(dev)] = True
            yield induced_subgraph(self.graph, sel[self.part.long()])
