"""Host-side mirror of the reference's `layers` package (layers/__init__.py:1-5): same class
names, constructor arguments, forward() signatures and state_dict keys; the arithmetic runs in
libgnm.so (HIP, gfx950).  Per-edge tensors at these module boundaries are in the caller's
edge-id order, as with DGL."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

import torch.nn.functional as F

from . import engine
from .graph import as_assembly_graph

__all__ = ["GatedGCN_1d", "GraphGatedGCN", "ScorePredictor", "NodeEncoder", "EdgeEncoder"]


KERNEL_WIDTHS = (32, 64, 128, 256)      # the hidden sizes the row kernels are instantiated for (GNM_DISPATCH_H)
# The widths the MODULES run on.  64 is left out since round 6: a 64-wide layer has only the round-1 schedule (generic GEMMs at K = 64,
# separate by-destination / by-source passes), and at the metric's graph that is SLOWER than the same model zero-padded to 128 on the
# fused kernels, the two-sided sweeps and the chained backward -- 222 against ~160 ms per step at L = 8, 50.7 against ~31 at
# BASELINE config 1's H = 64 / L = 1 (profiles/r06_h64_padded.txt) -- although the padded model moves twice the bytes.  32-wide layers
# stay native (141 against 144-145 ms at L = 8; a quarter of the memory).  GNM_NATIVE_64=1 / layers.RUN_WIDTHS = KERNEL_WIDTHS: the 64-wide kernels again (they stay built and tested).
RUN_WIDTHS = KERNEL_WIDTHS if os.environ.get("GNM_NATIVE_64", "0") == "1" else (32, 128, 256)
if os.environ.get("GNM_RUN_WIDTHS"):        # e.g. GNM_RUN_WIDTHS=128,256 (A/B runs): any subset of KERNEL_WIDTHS
    RUN_WIDTHS = tuple(sorted(int(w) for w in os.environ["GNM_RUN_WIDTHS"].split(",")))
    if not RUN_WIDTHS or any(w not in KERNEL_WIDTHS for w in RUN_WIDTHS):
        raise ValueError(f"GNM_RUN_WIDTHS={os.environ['GNM_RUN_WIDTHS']!r}: expected a subset of {KERNEL_WIDTHS}")


def padded_width(width: int) -> int:
    """The kernel width a layer of `width` output channels runs on: itself, or the next one up with dead channels.  Above the widest
    instantiation a layer runs as 256-column problems between full-width dense products (engine.WIDE_CHUNK: BatchNorm layers; slow
    by construction, but nn.Linear(in, out) of gated_gcn_full.py:44-50 takes any width and so does this)."""
    for w in RUN_WIDTHS:
        if width <= w:
            return w
    c = engine.WIDE_CHUNK
    return (width + c - 1) // c * c


def _params_of(module: nn.Module, prefix: str = ""):
    return {prefix + k: v for k, v in module.named_parameters()}


class _LayerFn(torch.autograd.Function):
    """One GatedGCN_1d.forward with edge-id-order e at the boundary."""

    @staticmethod
    def forward(ctx, graph, need, norm, residual, h, e, *flat):
        names = _LAYER_KEYS
        batch_norm, ln_width = norm           # ln_width: the layer's real out_channels (LayerNorm statistics, engine.layer_forward)
        P = {"gnn.convs.0." + k: v for k, v in zip(names, flat)}
        idx = graph.index(h.device)
        N, E, H = graph.num_nodes(), graph.num_edges(), flat[0].shape[0]        # H = out_channels (A_1.weight is [out, in])
        perm = idx["perm"].long()
        e_int = e.detach().index_select(0, perm).contiguous()
        prm = engine.layer_params(P, 0)
        h_int = engine.node_rows_in(idx, engine._f32c(h.detach()))       # caller's node numbering -> internal (graph.py)
        h_out, e_out, saved = engine.layer_forward(idx, N, E, H, prm, h_int, e_int, need, batch_norm, residual, ln_width=ln_width)
        ctx.graph, ctx.saved, ctx.P, ctx.dims, ctx.bn, ctx.res, ctx.lnw = graph, saved, P, (N, E, H), batch_norm, residual, ln_width
        out_e = torch.empty_like(e_out)
        out_e.index_copy_(0, perm, e_out)
        return engine.node_rows_out(idx, h_out), out_e

    @staticmethod
    def backward(ctx, gh_out, ge_out):
        if ctx.saved is None:
            raise RuntimeError("GatedGCN_1d: backward called twice (retain_graph is not supported) or forward ran without grad")
        N, E, H = ctx.dims
        idx = ctx.graph.index(gh_out.device)
        perm = idx["perm"].long()
        prm = engine.layer_params(ctx.P, 0)
        ge = ge_out.index_select(0, perm).contiguous()        # fresh buffer, overwritten below
        gh_in, ge_in, g = engine.layer_backward(idx, N, E, H, prm, ctx.saved,
                                                engine.node_rows_in(idx, engine._f32c(gh_out)), ge, ctx.bn,
                                                residual=ctx.res, ln_width=ctx.lnw)
        gh_in = engine.node_rows_out(idx, gh_in)
        ctx.saved = None
        ge_user = torch.empty_like(ge_in)
        ge_user.index_copy_(0, perm, ge_in)
        grads = []
        for j, k in enumerate(engine.LIN5):
            grads += [g["W5"][j * H:(j + 1) * H], g["b5"][j * H:(j + 1) * H]]
        grads += [g["W3"], g["b3"], g["gamma_h"], g["beta_h"], g["gamma_e"], g["beta_e"]]
        return (None, None, None, None, gh_in, ge_user) + tuple(grads)


_LAYER_KEYS = tuple(f"{k}.{w}" for k in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3") for w in ("weight", "bias")) + \
    ("bn_h.weight", "bn_h.bias", "bn_e.weight", "bn_e.bias")


class _Norm(nn.Module):
    """Parameter holder with the state_dict of BatchNorm1d(track_running_stats=False) /
    LayerNorm (weight, bias; no running buffers: gated_gcn_full.py:55-59).  Which of the two the
    layer applies is GatedGCN_1d.batch_norm."""

    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))
        self.eps = 1e-5


class GatedGCN_1d(nn.Module):
    """gated_gcn_full.py:8-157.  forward(g, h, e) -> (h, e)."""

    def __init__(self, in_channels, out_channels, batch_norm, dropout=0, residual=True):
        super().__init__()
        if not 0 <= dropout < 1:
            raise ValueError(f"GatedGCN_1d: dropout={dropout}")
        if not batch_norm and out_channels > KERNEL_WIDTHS[-1]:
            raise NotImplementedError(f"GatedGCN_1d(batch_norm=False) at width {out_channels}: the LayerNorm kernels hold a row in one "
                                      f"wavefront (widths up to {KERNEL_WIDTHS[-1]}); BatchNorm layers run at any width")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.dropout = dropout
        self.batch_norm = batch_norm
        # gated_gcn_full.py:41-42: a layer that changes the width silently drops the residual.  The model never builds
        # such a layer (processor.py:11-12); it runs on the generic GEMM route instead of the fused H = 128 kernels.
        self.residual = bool(residual) and in_channels == out_channels
        for k in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3"):     # creation order = the reference's
            setattr(self, k, nn.Linear(in_channels, out_channels))
        self.bn_h = _Norm(out_channels)
        self.bn_e = _Norm(out_channels)

    def forward(self, g, h, e):
        g = as_assembly_graph(g, h.device)
        P = dict(self.named_parameters())
        flat = [P[k] for k in _LAYER_KEYS]
        need = torch.is_grad_enabled() and any(t.requires_grad for t in [h, e] + flat)
        W, Wp = self.out_channels, padded_width(self.out_channels)
        if Wp != W:
            # run the next kernel width up with dead output channels (zero weight rows, bias 0, norm weight 1 / bias 0); with
            # the residual (in == out) the inputs get the same dead channels.  autograd slices everything back.
            d = Wp - W
            di = d if self.residual else 0
            flat = [F.pad(t, (0, di, 0, d)) if t.dim() == 2 else
                    F.pad(t, (0, d), value=1.0 if k in ("bn_h.weight", "bn_e.weight") else 0.0) for k, t in zip(_LAYER_KEYS, flat)]
            if di:
                h, e = F.pad(h, (0, di)), F.pad(e, (0, di))
        h, e = _LayerFn.apply(g, need, (bool(self.batch_norm), W), self.residual, h, e, *flat)
        if Wp != W:
            h, e = h[:, :W], e[:, :W]
        if self.dropout and self.training:
            # gated_gcn_full.py:154: dropout on the node output only, after the residual.  Never enabled by the model
            # (processor.py:12 passes no dropout), so it is not fused: the mask comes from torch's generator on h's device
            h = torch.nn.functional.dropout(h, self.dropout, training=True)
        return h, e


class GraphGatedGCN(nn.Module):
    """processor.py:8-20.  forward(graph, h, e) -> (h, e)."""

    def __init__(self, num_layers, hidden_features, batch_norm):
        super().__init__()
        self.convs = nn.ModuleList([
            GatedGCN_1d(hidden_features, hidden_features, batch_norm) for _ in range(num_layers)
        ])

    def forward(self, graph, h, e):
        graph = as_assembly_graph(graph, h.device)
        for conv in self.convs:
            h, e = conv(graph, h, e)
        return h, e


class _PredFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, need, x, e, W1, b1, W2, b2):
        idx = graph.index(x.device)
        N, E, H = graph.num_nodes(), graph.num_edges(), x.shape[1]
        perm = idx["perm"].long()
        e_int = e.detach().index_select(0, perm).contiguous()
        scores, saved = engine.predictor_forward(idx, N, E, H, W1.detach(), b1.detach(), W2.detach(), b2.detach(),
                                                 engine.node_rows_in(idx, engine._f32c(x.detach())), e_int, need)
        ctx.graph, ctx.saved, ctx.dims = graph, saved, (N, E, H)
        ctx.W1, ctx.W2 = W1.detach(), W2.detach()
        return scores

    @staticmethod
    def backward(ctx, gscores):
        if ctx.saved is None:      # the saved pre-activation is overwritten in place by its gradient
            raise RuntimeError("ScorePredictor: backward called twice (retain_graph is not supported) or forward ran without grad")
        N, E, H = ctx.dims
        idx = ctx.graph.index(gscores.device)
        perm = idx["perm"].long()
        gx, ge, g = engine.predictor_backward(idx, N, E, H, ctx.W1, ctx.W2, ctx.saved, gscores)
        gx = engine.node_rows_out(idx, gx)
        ctx.saved = None
        ge_user = torch.empty_like(ge)
        ge_user.index_copy_(0, perm, ge)
        return None, None, gx, ge_user, g["W1"], g["b1"], g["W2"], g["b2"]


class ScorePredictor(nn.Module):
    """score_predictor.py:5-25.  forward(graph, x, e) -> [E,1] in edge-id order."""

    def __init__(self, in_features, hidden_edge_scores):
        super().__init__()
        self.W1 = nn.Linear(3 * in_features, hidden_edge_scores)
        self.W2 = nn.Linear(hidden_edge_scores, 1)

    def forward(self, graph, x, e):
        graph = as_assembly_graph(graph, x.device)
        W1 = self.W1.weight
        H = x.shape[1]
        Hp = padded_width(H)
        if Hp != H:                           # dead input channels: zero columns in each of W1's three H-wide blocks
            x, e = F.pad(x, (0, Hp - H)), F.pad(e, (0, Hp - H))
            W1 = F.pad(W1.reshape(W1.shape[0], 3, H), (0, Hp - H)).reshape(W1.shape[0], 3 * Hp)
        ts = (x, e, W1, self.W1.bias, self.W2.weight, self.W2.bias)
        need = torch.is_grad_enabled() and any(t.requires_grad for t in ts)
        return _PredFn.apply(graph, need, *ts)


class NodeEncoder(nn.Module):
    """node_encoder.py:4-28 (unused by the model: full_graph.py:14); importable for parity."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x):
        return self.linear(x)


class EdgeEncoder(nn.Module):
    """edge_encoder.py:4-28 (unused by the model: full_graph.py:16); importable for parity."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x):
        return self.linear(x)
