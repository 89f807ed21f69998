#!/usr/bin/env python3
"""Golden vectors for the greedy decode step, produced by the REFERENCE's own
functions.  inference.py cannot be imported here (dgl, Bio are absent), so the four pure functions
this fixture needs are compiled from the reference file's AST at generation time (nothing of the
reference's text is written out): get_contig_length, walk_forwards, walk_backwards, sample_edges
(inference.py:20-77,270-277).  Adjacency dicts are built exactly as graph_parser.get_neighbors /
get_predecessors / get_edges do (graph_parser.py:13-73).  Run here only:  python tests/golden/make_golden_decode.py"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from gnnome_assembly_amd import synth  # noqa: E402

REF = "/root/reference/inference.py"
WANT = {"get_contig_length", "walk_forwards", "walk_backwards", "sample_edges"}


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    assert {n.name for n in body} == WANT
    ns = {"torch": torch}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


class FakeGraph:      # the two attributes get_contig_length reads (inference.py:25,27)
    def __init__(self, prefix, read_len):
        self.edata = {"prefix_length": torch.from_numpy(prefix)}
        self.ndata = {"read_length": torch.from_numpy(read_len)}


def main():
    ref = reference_functions()
    rng = np.random.default_rng(11)
    src, dst, n = synth.make_graph(300, seed=4, permute_edge_ids=True)
    e = src.size
    # a few duplicate edges and a dead end region, as real graphs have
    dup = rng.choice(e, 12, replace=False)
    src = np.concatenate((src, src[dup])).astype(np.int32)
    dst = np.concatenate((dst, dst[dup])).astype(np.int32)
    e = src.size
    scores = (rng.standard_normal(e) * 2.0).astype(np.float32)
    prefix = rng.integers(500, 12000, e).astype(np.int64)
    read_len = rng.integers(8000, 25000, n).astype(np.int64)
    succs = {i: [] for i in range(n)}
    preds = {i: [] for i in range(n)}
    edges = {}
    for k, (s, d) in enumerate(zip(src.tolist(), dst.tolist())):
        succs[s].append(d)
        preds[d].append(s)
        edges[(s, d)] = k
    g = FakeGraph(prefix, read_len)
    p = torch.from_numpy(scores)
    K = 40
    starts = rng.choice(e, K, replace=False)
    old_masks = np.zeros((K, n), bool)
    walks, offs, vis_f, vis_b, lens = [], [0], np.zeros((K, n), bool), np.zeros((K, n), bool), []
    for i, k in enumerate(starts):
        if i % 3:       # two thirds of the cases start from a partly consumed graph
            blk = rng.integers(0, n - 40)
            old_masks[i, blk:blk + rng.integers(5, 40)] = True
            old_masks[i, rng.choice(n, 10, replace=False)] = True
        old = set(np.flatnonzero(old_masks[i]).tolist())
        s0, d0 = int(src[k]), int(dst[k])
        wf, vf = ref["walk_forwards"](d0, p, succs, preds, edges, old)
        wb, vb = ref["walk_backwards"](s0, p, preds, succs, edges, old | vf)
        walk = wb + wf
        walks.extend(walk)
        offs.append(len(walks))
        vis_f[i, list(vf)] = True
        vis_b[i, list(vb)] = True
        lens.append(int(ref["get_contig_length"](walk, g, edges)))
    torch.manual_seed(123)
    sub_scores = torch.from_numpy(scores[:1000].copy())
    idx = ref["sample_edges"](sub_scores, 50).numpy()
    out = os.path.join(HERE, "decode_walks.npz")
    np.savez_compressed(out, src=src, dst=dst, n=n, scores=scores, prefix_length=prefix, read_length=read_len,
                        starts=starts.astype(np.int64), visited_old=np.packbits(old_masks, axis=1),
                        walks=np.asarray(walks, np.int32), walk_offsets=np.asarray(offs, np.int64),
                        visited_f=np.packbits(vis_f, axis=1), visited_b=np.packbits(vis_b, axis=1),
                        contig_length=np.asarray(lens, np.int64), sample_seed=123, sample_nb_paths=50,
                        sample_scores=sub_scores.numpy(), sample_idx=idx.astype(np.int64))
    print(out, os.path.getsize(out), "bytes;", K, "walks, mean length", len(walks) / K)


if __name__ == "__main__":
    main()
