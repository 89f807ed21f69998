"""CPU restatement of the reference's greedy decode -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product path is gnnome_assembly_amd/decode.py (C++ walks in
libgnm.so).  Each function cites the reference lines it follows.

Pinning: walk_forwards / walk_backwards / get_contig_length / sample_edges are checked against
tests/golden/decode_walks.npz, produced by the reference's own functions (tests/golden/make_golden_decode.py).
get_contigs as a whole is NOT pinned at the DGL boundary (dgl.remove_self_loop, dgl.node_subgraph are
absent here): the loop below restates inference.py:182-267 with the documented semantics of those two
calls (induced sub-graph of the unvisited nodes, edges in ascending edge id, self loops dropped from
the candidates)."""
from __future__ import annotations

import numpy as np
import torch


def build_adjacency(src, dst, n):
    """graph_parser.get_neighbors / get_predecessors / get_edges (graph_parser.py:13-73): per-node
    successor and predecessor lists in edge-id order, (src, dst) -> edge id (the LAST id for duplicates)."""
    succs = {v: [] for v in range(int(n))}
    preds = {v: [] for v in range(int(n))}
    edges = {}
    for k, (s, d) in enumerate(zip(np.asarray(src).tolist(), np.asarray(dst).tolist())):
        succs[s].append(d)
        preds[d].append(s)
        edges[(s, d)] = k
    return succs, preds, edges


def _greedy(start, scores, nbrs, edge_of, visited_old):
    """Common body of inference.py:31-52 and :55-76.  `edge_of(current, n)` gives the edge id."""
    cur, walk, seen = int(start), [], set()
    while True:
        walk.append(cur)
        seen.add(cur)
        seen.add(cur ^ 1)                       # the reverse-complement read is consumed too (:38-39)
        options = nbrs[cur]
        if not options:
            break
        if len(options) == 1:                   # forced move, taken even into visited territory (:42-44)
            cur = options[0]
            continue
        free = [v for v in options if v not in visited_old and v not in seen]
        if not free:
            break
        p = scores[[edge_of(cur, v) for v in free]]
        cur = free[int(torch.argmax(p))]        # topk(k=1): the best-scored free neighbour (:49-51)
    return walk, seen


def walk_forwards(start, scores, succs, edges, visited_old):
    """inference.py:31-52"""
    return _greedy(start, scores, succs, lambda a, b: edges[(a, b)], visited_old)


def walk_backwards(start, scores, preds, edges, visited_old):
    """inference.py:55-77 (the walk is returned in forward orientation)"""
    w, seen = _greedy(start, scores, preds, lambda a, b: edges[(b, a)], visited_old)
    return w[::-1], seen


def get_contig_length(walk, prefix_length, read_length, edges):
    """inference.py:20-28: prefixes of all overlaps on the walk + the whole last read"""
    total = 0
    for a, b in zip(walk[:-1], walk[1:]):
        total += int(prefix_length[edges[(a, b)]])
    return total + int(read_length[walk[-1]])


def sample_edges(edge_scores, nb_paths):
    """inference.py:270-277: nb_paths independent draws from softmax-free p ~ sigmoid(score)"""
    p = torch.sigmoid(edge_scores).squeeze()
    p = p.masked_fill(p < 1e-9, 1e-9)
    p = p / p.sum()
    return torch.distributions.categorical.Categorical(p.repeat(nb_paths, 1)).sample()


def get_contigs(src, dst, n, scores, prefix_length, read_length, nb_paths=50, len_threshold=20, sampler=sample_edges):
    """inference.py:182-253 with get_subgraph (:256-267)."""
    src, dst = np.asarray(src), np.asarray(dst)
    succs, preds, edges = build_adjacency(src, dst, n)
    scores = torch.as_tensor(scores)
    visited, contigs = set(), []
    while True:
        keep = np.ones(int(n), bool)
        keep[list(visited)] = False
        eid = np.flatnonzero(keep[src] & keep[dst] & (src != dst))
        if eid.size == 0:
            break
        picks = sampler(scores[eid], nb_paths)
        walks, seens = [], []
        for i in picks.tolist():
            s0, d0 = int(src[eid[i]]), int(dst[eid[i]])
            wf, vf = walk_forwards(d0, scores, succs, edges, visited)
            wb, vb = walk_backwards(s0, scores, preds, edges, visited | vf)
            walks.append(wb + wf)
            seens.append(vf | vb)
        lens = [get_contig_length(w, prefix_length, read_length, edges) for w in walks]
        best = int(np.argmax(lens))                   # max(): the first of equally long walks (:225-226)
        walk, seen = walks[best], set(seens[best])
        for a, b in zip(walk[:-1], walk[1:]):         # nodes the walk jumped over (:231-239)
            t = set(succs[a]) & set(preds[b])
            seen |= t | {v ^ 1 for v in t}
        if len(walk) < len_threshold:
            break
        contigs.append(walk)
        visited |= seen
    return contigs
