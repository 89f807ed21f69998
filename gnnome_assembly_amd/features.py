"""GPU counterparts of the reference's input preparation (utils.preprocess_graph utils.py:67-74,
utils.add_positional_encoding utils.py:97-138, and the per-step assembly train.py:245-251):
features are computed once on the device from the graph index and stay resident."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .engine import _call, _ptr, _stream, on_device_of, scratch

__all__ = ["positional_encoding", "edge_features", "prepare_graph"]


@on_device_of(lambda graph, *a, **k: graph.device)
def positional_encoding(graph, pe_dim: int = 16, alpha: float = 0.95) -> torch.Tensor:
    """[N, 2+pe_dim] fp32 = in_deg | out_deg | k-step PageRank (the `pe` argument of the model)."""
    lib = _lib.load()
    dev = graph.device
    if dev.type != "cuda":
        raise _lib.GnmError("positional_encoding: the graph must be on a HIP device")
    idx = graph.index(dev)
    N, E = graph.num_nodes(), graph.num_edges()
    pe = torch.empty(N, pe_dim + 2, dtype=torch.float32, device=dev)
    need = lib.gnm_pagerank_pe_workspace_bytes(N)
    ws = scratch(dev).ws(need)
    _call("gnm_pagerank_pe", N, E, _ptr(idx["isrc"]), _ptr(idx["in_ptr"]), _ptr(idx["out_ptr"]), pe_dim,
          C.c_double(alpha), _ptr(pe), _ptr(ws), need, _stream())
    from .engine import node_rows_out
    return node_rows_out(idx, pe)      # the index may use an internal node numbering (graph.py); features leave in the caller's


@on_device_of(lambda overlap_length, overlap_similarity: overlap_similarity)
def edge_features(overlap_length: torch.Tensor, overlap_similarity: torch.Tensor) -> torch.Tensor:
    """[E,2] fp32 z-scored edge features in the caller's edge-id order (the `e` argument)."""
    dev = overlap_similarity.device
    if dev.type != "cuda":
        raise _lib.GnmError("edge_features: tensors must be on a HIP device")
    a = overlap_length.to(torch.float32).contiguous()
    b = overlap_similarity.to(torch.float32).contiguous()
    E = a.numel()
    e = torch.empty(E, 2, dtype=torch.float32, device=dev)
    sc = scratch(dev)
    _call("gnm_edge_feats_zscore", E, _ptr(a), _ptr(b), _ptr(e), _ptr(sc.partials), sc.partials.numel() * 8, _stream())
    return e


def prepare_graph(graph, nb_pos_enc: int = 16):
    """preprocess_graph + add_positional_encoding on the device: fills graph.ndata['x','pe','in_deg',
    'out_deg'] and graph.edata['e'] (from edata 'overlap_length' / 'overlap_similarity') and returns
    (x, e, pe18) exactly as train.py:246-251 hands them to the model."""
    pe18 = positional_encoding(graph, nb_pos_enc)
    graph.ndata["in_deg"], graph.ndata["out_deg"], graph.ndata["pe"] = pe18[:, 0], pe18[:, 1], pe18[:, 2:]
    graph.ndata["x"] = torch.ones(graph.num_nodes(), 1, device=graph.device)        # utils.py:69
    e = None
    if "overlap_length" in graph.edata and "overlap_similarity" in graph.edata:
        e = edge_features(graph.edata["overlap_length"], graph.edata["overlap_similarity"])
        graph.edata["e"] = e
    return graph.ndata["x"], e, pe18
