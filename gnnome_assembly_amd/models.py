"""Host-side mirror of the reference's `models` package (models/full_graph.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

import torch.nn.functional as F

from . import dp, engine, layers
from .graph import as_assembly_graph

__all__ = ["GraphGatedGCNModel", "BCEWithLogitsLoss", "flatten_parameters"]


def _engine_order(names):
    """Parameter names in the order the kernels like them in memory: per layer the five stacked projection weights
    (A_1 A_2 A_3 B_1 B_2: one [5H,H] operand), their five biases ([5H]), then the rest; encoders and predictor last."""
    rank = {}
    for k in names:
        parts = k.split(".")
        if parts[0] == "gnn":
            layer, mod, kind = int(parts[2]), parts[3], parts[4]
            if mod in engine.LIN5:
                rank[k] = (0, layer, 0 if kind == "weight" else 1, engine.LIN5.index(mod))
            else:
                rank[k] = (0, layer, 2, names.index(k))
        else:
            rank[k] = (1, 0, 0, names.index(k))
    return sorted(names, key=lambda k: rank[k])


def flatten_parameters(model: nn.Module) -> torch.Tensor:
    """Move every parameter of `model` into ONE contiguous fp32 buffer (values preserved; each parameter becomes a
    view of it) laid out in _engine_order, so that the engine takes [5H,H] / [5H] views of the stacked projection
    parameters instead of torch.cat-ing them on every pass, and dp.FlatGradients (which follows the same order)
    can hand the kernels gradient targets of the same shape.  Returns the buffer.  state_dict keys, optimizers
    and load_state_dict are unaffected; model.to(device) un-flattens (the next forward flattens again)."""
    named = dict(model.named_parameters())
    order = _engine_order(list(named))
    dev = next(iter(named.values())).device
    flat = torch.empty(sum(p.numel() for p in named.values()), dtype=torch.float32, device=dev)
    o = 0
    with torch.no_grad():
        for slot, k in enumerate(order):
            p = named[k]
            v = flat[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            p._gnm_slot = slot
            o += p.numel()
    model._gnm_flat = flat
    return flat


def _is_flat(model: nn.Module) -> bool:
    flat = getattr(model, "_gnm_flat", None)
    if flat is None:
        return False
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    return all(p.device == flat.device and lo <= p.data_ptr() < hi for p in model.parameters())


class _ModelFn(torch.autograd.Function):
    """Whole-model forward/backward: the edge stream stays in internal order from the encoder
    to the predictor, activations are released layer by layer in backward."""

    @staticmethod
    def forward(ctx, graph, e, pe, num_layers, names, need, norm, *flat):
        # `need` (grad mode on and some parameter requires grad) is decided by the caller: inside
        # Function.forward grad mode is always off, and needs_input_grad stays set under no_grad.
        # norm = (batch_norm, real hidden width): LayerNorm statistics run over the model's real width when the kernels
        # run it zero-padded to the next width up (gated_gcn_full.py:58-59: nn.LayerNorm(out_channels))
        batch_norm, ln_width = norm
        P = {k: v.detach() for k, v in zip(names, flat)}
        scores, saved = engine.model_forward(graph, e.detach(), pe.detach(), P, num_layers, need, batch_norm, ln_width=ln_width)
        ctx.graph, ctx.saved, ctx.P, ctx.names, ctx.L, ctx.bn, ctx.lnw = graph, saved, P, names, num_layers, batch_norm, ln_width
        ctx.params = flat if need else None
        return scores

    @staticmethod
    def backward(ctx, gscores):
        if ctx.saved is None:
            raise RuntimeError("GraphGatedGCNModel: backward called twice or forward ran without grad")
        # Fast path: every .grad is a view of a dp.FlatGradients buffer that has been zeroed since its last use
        # (the training loops call flat.zero_() before each backward) -> the kernels write the gradients straight
        # into those views and autograd is told "no gradient": accumulating x into zeros is x, and the ~140
        # accumulation kernels per step disappear.  Anything else takes the ordinary autograd route.
        fg = dp.fresh_flat_gradients(ctx.params)
        if fg is not None and all(ctx.needs_input_grad[7:]):
            out = {k: p.grad for k, p in zip(ctx.names, ctx.params)}
            engine.model_backward(ctx.graph, ctx.P, ctx.L, ctx.saved, gscores, ctx.bn, out=out, ln_width=ctx.lnw)
            fg.fresh = False
            ctx.saved = None
            return (None,) * (7 + len(ctx.names))
        G = engine.model_backward(ctx.graph, ctx.P, ctx.L, ctx.saved, gscores, ctx.bn, ln_width=ctx.lnw)
        ctx.saved = None
        return (None, None, None, None, None, None, None) + tuple(G[k] for k in ctx.names)


def _pad_param(name: str, v: torch.Tensor, H: int, Hp: int) -> torch.Tensor:
    """One parameter of a width-H model as the parameter of the width-Hp model whose extra channels are dead."""
    d = Hp - H
    leaf = name.split(".")[-2] if "." in name else name
    if name.startswith("predictor.W1.weight"):                     # [hs, 3H]: three H-wide blocks (x[src] | x[dst] | e)
        return F.pad(v.reshape(v.shape[0], 3, H), (0, d)).reshape(v.shape[0], 3 * Hp)
    if name.startswith(("predictor.", "linear1_edge.")):
        return v
    if name.startswith(("linear_pe.", "linear2_edge.")):           # [H, in] / [H]: new output rows
        return F.pad(v, (0, 0, 0, d)) if v.dim() == 2 else F.pad(v, (0, d))
    if leaf in ("bn_h", "bn_e"):
        return F.pad(v, (0, d), value=1.0 if name.endswith("weight") else 0.0)
    return F.pad(v, (0, d, 0, d)) if v.dim() == 2 else F.pad(v, (0, d))     # [H, H] layer weights, [H] biases


class GraphGatedGCNModel(nn.Module):
    """models/full_graph.py:11-29.  forward(graph, x, e, pe) -> scores [E,1] (edge-id order).

    `graph` is an AssemblyGraph (or anything AssemblyGraph-compatible on a HIP device); `x` is
    ignored exactly as in the reference (full_graph.py:23 overwrites it)."""

    def __init__(self, node_features, edge_features, hidden_features, hidden_edge_features, num_layers,
                 hidden_edge_scores, batch_norm, nb_pos_enc):
        super().__init__()
        self.linear_pe = nn.Linear(nb_pos_enc + 2, hidden_features)
        self.linear1_edge = nn.Linear(edge_features, hidden_edge_features)
        self.linear2_edge = nn.Linear(hidden_edge_features, hidden_features)
        self.gnn = layers.GraphGatedGCN(num_layers, hidden_features, batch_norm)
        self.predictor = layers.ScorePredictor(hidden_features, hidden_edge_scores)
        self.num_layers = num_layers
        self.batch_norm = bool(batch_norm)

    def flatten_parameters(self) -> torch.Tensor:
        """See models.flatten_parameters.  Call it after .to(device) and BEFORE building dp.FlatGradients so that the
        gradient buffer follows the same layout (forward() does it on its own otherwise, but a FlatGradients made
        earlier then keeps state_dict order and the kernels' stacked gradients are copied instead of written in place)."""
        return flatten_parameters(self)

    def forward(self, graph, x, e, pe):
        graph = as_assembly_graph(graph, pe.device)       # a DGLGraph(-like) object is wrapped once and cached on itself
        H = self.linear_pe.out_features
        Hp = layers.padded_width(H)
        if Hp != H:
            # a width the kernels are not built for (they are for 32 / 64 / 128 / 256): run the next one up with zero-padded
            # parameters -- the extra channels stay exactly zero through every layer (t = 0 -> bn -> relu -> 0; their gates
            # are 0.5 and gate zeros) and autograd slices the gradients back out of the padded tensors
            names, flat = zip(*self.named_parameters())
            padded = tuple(_pad_param(k, v, H, Hp) for k, v in zip(names, flat))
            need = torch.is_grad_enabled() and any(p.requires_grad for p in flat)
            return _ModelFn.apply(graph, e, pe, self.num_layers, names, need, (self.batch_norm, H), *padded)
        if pe.is_cuda and not _is_flat(self):
            flatten_parameters(self)          # once per device placement: stacked-parameter views instead of torch.cat
        names, flat = zip(*self.named_parameters())
        need = torch.is_grad_enabled() and any(p.requires_grad for p in flat)
        if torch.is_grad_enabled() and (e.requires_grad or pe.requires_grad):
            # the reference never differentiates its inputs (train.py:245-258); the whole-model backward
            # stops at the encoders, so refuse instead of returning a silent None for these gradients
            raise NotImplementedError("GraphGatedGCNModel: gradients w.r.t. the inputs e / pe are not computed; "
                                      "detach them (the stand-alone layers do return input gradients)")
        return _ModelFn.apply(graph, e, pe, self.num_layers, names, need, (self.batch_norm, H), *flat)


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, y, pos_weight):
        loss, gs = engine.bce_with_logits(scores.detach(), y, pos_weight)
        ctx.gs = gs.reshape(scores.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        return ctx.gs * gl, None, None


class BCEWithLogitsLoss(nn.Module):
    """torch.nn.BCEWithLogitsLoss(pos_weight=[pw]) with mean reduction as train.py:210-211 uses
    it, as one fused HIP pass that also produces d loss / d logits."""

    def __init__(self, pos_weight):
        super().__init__()
        self.pos_weight = float(pos_weight.reshape(-1)[0]) if torch.is_tensor(pos_weight) else float(pos_weight)

    def forward(self, scores, y):
        return _BCEFn.apply(scores, y, self.pos_weight)
