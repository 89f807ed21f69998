"""Seeded synthetic assembly (overlap) graphs, shaped like the reference's Raven graphs.

The reference has no generator of its own (graphs come from Raven, an external
binary: graph_dataset.py:120).  This follows SURVEY.md section 8(d):

* ``R`` reads sorted by start position; node ``2i`` is the + strand of read ``i`` and
  node ``2i+1`` its reverse complement (the ``v ^ 1`` convention of
  algorithms.py:139 / inference.py:39).
* read ``i`` overlaps the next ``k_i = 1 + Poisson(mean_degree-1)`` reads (clipped to
  16): edge ``2i -> 2j`` plus the strand mirror ``(2j+1) -> (2i+1)``.
* 0.5 % extra long-range "repeat" edges (with mirrors), no self loops.
* edge ids are assigned src-major (what dgl.from_networkx does, graph_parser.py:297),
  optionally followed by a seeded random permutation of the ids.

Feature distributions follow utils.py:67-74 (z-scored overlap length / similarity),
utils.py:102-103 (float degrees) and utils.py:122-138 (16-step PageRank, alpha=0.95).
Everything is numpy; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_graph", "pagerank_pe", "make_inputs", "CHR_SCALE"]

# relative sizes (evaluate.py:10,28-30 chromosome lengths, chr19 = 1.0)
CHR_SCALE = {"chr19": 1.0, "chr20": 1.073, "chr21": 0.731, "chr1": 4.03}


def make_graph(reads: int, seed: int = 0, mean_degree: float = 5.0, max_degree: int = 16,
               long_range_frac: float = 0.005, permute_edge_ids: bool = False):
    """Return (src, dst) int32 arrays in edge-id order; N = 2*reads nodes."""
    rng = np.random.default_rng(seed)
    R = int(reads)
    k = 1 + rng.poisson(mean_degree - 1.0, size=R)
    k = np.minimum(k, max_degree)
    k = np.minimum(k, np.maximum(R - 1 - np.arange(R), 0))  # no wrap at the chromosome end
    i = np.repeat(np.arange(R, dtype=np.int64), k)
    first = np.cumsum(k) - k
    off = np.arange(i.size, dtype=np.int64) - np.repeat(first, k) + 1
    j = i + off
    n_long = int(long_range_frac * i.size)
    if n_long > 0 and R > 2:
        li = rng.integers(0, R, size=n_long)
        lj = rng.integers(0, R, size=n_long)
        keep = li != lj
        i = np.concatenate([i, li[keep]])
        j = np.concatenate([j, lj[keep]])
    # forward strand edge 2i -> 2j and its mirror (2j+1) -> (2i+1)
    src = np.concatenate([2 * i, 2 * j + 1])
    dst = np.concatenate([2 * j, 2 * i + 1])
    # src-major edge ids (stable: keeps generation order inside one source node)
    order = np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
    if permute_edge_ids:
        p = rng.permutation(src.size)
        src, dst = src[p], dst[p]
    return src.astype(np.int32), dst.astype(np.int32), 2 * R


def pagerank_pe(src: np.ndarray, dst: np.ndarray, n: int, pe_dim: int = 16, alpha: float = 0.95):
    """k-step PageRank features, the arithmetic of utils.py:122-138 without scipy.

    P = (D^-1 A)^T, x <- alpha * P x + (1-alpha)/n ; float64 iterate, float32 features.
    """
    out_deg = np.bincount(src, minlength=n).astype(np.float64)
    dinv = 1.0 / (out_deg + 1e-9)
    dinv[out_deg < 1e-9] = 0.0
    x = np.full(n, 1.0 / n)
    w = dinv[src]
    pe = np.empty((n, pe_dim), dtype=np.float32)
    for t in range(pe_dim):
        x = alpha * np.bincount(dst, weights=w * x[src], minlength=n) + (1.0 - alpha) / n
        pe[:, t] = x.astype(np.float32)
    return pe


def make_inputs(src: np.ndarray, dst: np.ndarray, n: int, seed: int = 0, nb_pos_enc: int = 16,
                pos_frac: float = 0.8):
    """Model inputs the way train.py:245-251 assembles them.

    Returns dict(x[N,1], e[E,2], pe[N,nb_pos_enc+2], y[E], pos_weight).
    """
    rng = np.random.default_rng(seed + 1000003)
    E = src.size
    e = rng.standard_normal((E, 2)).astype(np.float32)
    e = ((e - e.mean(0)) / e.std(0, ddof=1)).astype(np.float32)  # torch .std() is unbiased
    in_deg = np.bincount(dst, minlength=n).astype(np.float32)
    out_deg = np.bincount(src, minlength=n).astype(np.float32)
    pe = pagerank_pe(src, dst, n, nb_pos_enc)
    pe = np.concatenate([in_deg[:, None], out_deg[:, None], pe], axis=1).astype(np.float32)
    y = (rng.random(E) < pos_frac).astype(np.float32)
    pos = float((y == 1).sum())
    neg = float((y == 0).sum())
    ratio = pos / max(neg, 1.0)          # train.py:181
    pos_weight = 1.0 / ratio if ratio > 0 else 1.0   # train.py:210
    x = np.ones((n, 1), dtype=np.float32)  # utils.py:69 (dead input of the model)
    return {"x": x, "e": e, "pe": pe, "y": y, "pos_weight": np.float32(pos_weight)}


def synth_state_dict(hidden_features: int, num_layers: int, seed: int = 0, nb_pos_enc: int = 16,
                     edge_features: int = 2, hidden_edge_features: int = 16,
                     hidden_edge_scores: int = 64, randomize_norm: bool = True):
    """Version-independent random parameters with the reference's key schema.

    Keys / shapes are GraphGatedGCNModel.state_dict()'s (full_graph.py:12-20,
    gated_gcn_full.py:44-59, score_predictor.py:8-9).  Linear weights and biases are
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the nn.Linear default distribution) drawn from a
    numpy PCG64 stream, so golden fixtures only need to store the seed.  Norm weights /
    biases are perturbed away from (1, 0) when ``randomize_norm`` so that gamma and beta
    paths are exercised.  Returns {key: float32 ndarray}.
    """
    rng = np.random.default_rng(seed + 7919)
    H = hidden_features
    sd = {}

    def lin(name, fan_in, fan_out):
        b = 1.0 / np.sqrt(fan_in)
        sd[name + ".weight"] = rng.uniform(-b, b, size=(fan_out, fan_in)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-b, b, size=(fan_out,)).astype(np.float32)

    lin("linear_pe", nb_pos_enc + 2, H)
    lin("linear1_edge", edge_features, hidden_edge_features)
    lin("linear2_edge", hidden_edge_features, H)
    for i in range(num_layers):
        for k in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3"):
            lin(f"gnn.convs.{i}.{k}", H, H)
        for k in ("bn_h", "bn_e"):
            if randomize_norm:
                sd[f"gnn.convs.{i}.{k}.weight"] = rng.uniform(0.5, 1.5, size=H).astype(np.float32)
                sd[f"gnn.convs.{i}.{k}.bias"] = rng.uniform(-0.3, 0.3, size=H).astype(np.float32)
            else:
                sd[f"gnn.convs.{i}.{k}.weight"] = np.ones(H, dtype=np.float32)
                sd[f"gnn.convs.{i}.{k}.bias"] = np.zeros(H, dtype=np.float32)
    lin("predictor.W1", 3 * H, hidden_edge_scores)
    lin("predictor.W2", hidden_edge_scores, 1)
    return sd


def tiny_edge_case_graph(seed: int = 0, n: int = 64, e: int = 256):
    """Small adversarial graph: isolated nodes, zero in-degree / zero out-degree nodes,
    self loops, duplicate edges, one high-degree hub, random edge-id order."""
    rng = np.random.default_rng(seed + 31)
    src = rng.integers(0, n - 8, size=e)          # nodes n-8.. are never sources
    dst = rng.integers(4, n - 4, size=e)          # nodes 0..3 have zero in-degree; n-4.. isolated
    src[:6] = dst[:6]                              # self loops
    src[6:12], dst[6:12] = src[12:18], dst[12:18]  # duplicate edges
    dst[20:60] = 17                                # hub with in-degree >= 40
    src[60:90] = 23                                # hub with out-degree >= 30
    return src.astype(np.int32), dst.astype(np.int32), n
