"""Greedy decode of contigs from edge scores: counterpart of `inference.get_contigs` with
`sample_edges`, `walk_forwards`, `walk_backwards`, `get_contig_length`, `get_subgraph`
(inference.py:20-77,182-277).

The step consumes `g.edata['score']` BY EDGE ID (inference.py:454) -- the reason the model returns
logits in the caller's edge-id order.  It is sequential, branchy CPU work (greedy walks over
adjacency lists); the reference runs it in Python over dict-of-lists, here the walks, the best-walk
choice and the visited-set update of one iteration are one C++ call on CSR arrays
(`gnm_decode_iteration`, libgnm.so, host side).  Python keeps the loop, the candidate edge list of the
not-yet-visited sub-graph and the sampling, which uses the reference's own torch call so that a
seeded run draws the same start edges.

Differences from the reference, both deliberate:
  * self loops are dropped from the CANDIDATES only, edge ids are never renumbered (the reference
    calls dgl.remove_self_loop and then indexes the renumbered scores with ids of the original
    graph, which is only consistent when there are no self loops);
  * a cycle of forced single-successor moves raises instead of looping forever."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib

__all__ = ["DecodeGraph", "sample_edges", "get_contigs", "infer_contigs"]


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class DecodeGraph:
    """Adjacency of a graph in edge-id order (what the reference pickles as *_succ.pkl, *_pred.pkl,
    *_edges.pkl; graph_parser.py:13-73), as CSR arrays."""

    def __init__(self, src, dst, num_nodes: int):
        self.src = np.ascontiguousarray(np.asarray(src), dtype=np.int32)
        self.dst = np.ascontiguousarray(np.asarray(dst), dtype=np.int32)
        self.n = int(num_nodes)
        e = self.src.size
        mk = lambda m: np.empty(m, np.int32)  # noqa: E731
        self.succ = (mk(self.n + 1), mk(e), mk(e))
        self.pred = (mk(self.n + 1), mk(e), mk(e))
        lib = _lib.load()
        _lib.check(lib.gnm_decode_build_adjacency(_p(self.src), _p(self.dst), self.n, e, *[_p(a) for a in self.succ],
                                                  *[_p(a) for a in self.pred]), "gnm_decode_build_adjacency")

    def successors(self, v: int) -> List[int]:
        ptr, nbr, _ = self.succ
        return nbr[ptr[v]:ptr[v + 1]].tolist()

    def predecessors(self, v: int) -> List[int]:
        ptr, nbr, _ = self.pred
        return nbr[ptr[v]:ptr[v + 1]].tolist()


def sample_edges(edge_scores: torch.Tensor, nb_paths: int) -> torch.Tensor:
    """inference.py:270-277: nb_paths independent start edges with p ~ sigmoid(score).  Up to 5e7
    probabilities this is the reference's own call (same draws for the same torch seed); beyond that
    its nb_paths-fold copy of the probability vector is replaced by torch.multinomial with replacement
    -- the same distribution without the copy."""
    p = torch.sigmoid(edge_scores.detach().float().cpu()).reshape(-1)
    p = p.masked_fill(p < 1e-9, 1e-9)
    p = p / p.sum()
    if p.numel() * nb_paths <= 50_000_000:
        return torch.distributions.categorical.Categorical(p.repeat(nb_paths, 1)).sample()
    return torch.multinomial(p, nb_paths, replacement=True)


def get_contigs(graph: DecodeGraph, scores, prefix_length, read_length, nb_paths: int = 50, len_threshold: int = 20,
                sampler: Callable[[torch.Tensor, int], torch.Tensor] = sample_edges,
                visited: Optional[np.ndarray] = None) -> List[List[int]]:
    """Iteratively extract walks until the best candidate has fewer than `len_threshold` nodes
    (inference.py:182-253).  `scores` [E] in edge-id order (logits), `prefix_length` [E], `read_length` [N]
    (g.edata['prefix_length'], g.ndata['read_length']).  Returns the walks as lists of node ids."""
    lib = _lib.load()
    n, e = graph.n, graph.src.size
    sc = np.ascontiguousarray(torch.as_tensor(scores).detach().float().cpu().numpy().reshape(-1))
    pl = np.ascontiguousarray(torch.as_tensor(prefix_length).cpu().numpy().reshape(-1), dtype=np.int64)
    rl = np.ascontiguousarray(torch.as_tensor(read_length).cpu().numpy().reshape(-1), dtype=np.int64)
    if sc.size != e or pl.size != e or rl.size != n:
        raise ValueError("scores / prefix_length need one entry per edge, read_length one per node")
    vis = np.zeros(n, np.uint8) if visited is None else visited
    if vis.dtype != np.uint8 or vis.size != n or not vis.flags.c_contiguous:
        raise ValueError("visited must be a contiguous uint8 array with one entry per node")
    sct = torch.from_numpy(sc)
    no_loop = graph.src != graph.dst
    walk = np.empty(2 * n + 2, np.int32)
    best_len = C.c_int64(0)
    contigs: List[List[int]] = []
    while True:
        free = vis == 0
        eid = np.flatnonzero(free[graph.src] & free[graph.dst] & no_loop)      # get_subgraph (:256-267)
        if eid.size == 0:
            break
        picks = eid[sampler(sct[torch.from_numpy(eid)], nb_paths).numpy().reshape(-1)]
        s0 = np.ascontiguousarray(graph.src[picks])
        d0 = np.ascontiguousarray(graph.dst[picks])
        length = lib.gnm_decode_iteration(n, _p(sc), _p(pl), _p(rl), *[_p(a) for a in graph.succ],
                                          *[_p(a) for a in graph.pred], _p(vis), int(picks.size), _p(s0), _p(d0),
                                          int(len_threshold), _p(walk), walk.size, C.byref(best_len))
        if length < 0:
            _lib.check(int(length), "gnm_decode_iteration")
        if length < len_threshold:
            break
        contigs.append(walk[:length].tolist())
    return contigs


def infer_contigs(model, graph, e, pe, prefix_length, read_length, nb_paths: int = 50, len_threshold: int = 20):
    """The per-graph body of inference.inference (inference.py:444-490): logits of the whole graph under
    no_grad in eval mode, stored by edge id, then greedy decode.  Returns (scores [E], walks)."""
    model.eval()
    with torch.no_grad():
        scores = model(graph, None, e, pe).squeeze(-1)                 # inference.py:453-454
    src, dst = graph.edges()
    dg = DecodeGraph(src.cpu().numpy(), dst.cpu().numpy(), graph.num_nodes())
    return scores, get_contigs(dg, scores, prefix_length, read_length, nb_paths, len_threshold)
