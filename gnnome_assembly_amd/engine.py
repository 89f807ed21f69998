"""Host-side orchestration of the HIP kernels (libgnm.so) for the GatedGCN edge-logit path.

Everything here is plumbing: torch owns the memory and the stream, every arithmetic step is a
C-ABI call (include/gnm.h).  There is no torch / CPU fallback: tensors must live on a HIP
device and the library must load, otherwise the calls raise.

Layout: all [E,*] tensors inside the layer stack are in the graph's internal (destination
sorted) edge order; `model_forward` gathers e_raw into that order and scatters the scores
back to the caller's edge-id order.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import os
import threading
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _lib

NT, NN, TN = 0, 1, 2
_D_FUSED = True    # Options.FUSED: use the W-stationary fused MFMA kernels where they exist (H = 128); tests flip it
EPS_BN = 1e-5   # nn.BatchNorm1d default (gated_gcn_full.py:55-56)

LIN5 = ("A_1", "A_2", "A_3", "B_1", "B_2")

# What a training forward keeps for the backward pass:
#   "saved"  every intermediate the backward kernels read: per layer P [N,5H], t [E,H], e_out [E,H], hf, inv_f, hb,
#            inv_b, z [N,H] (137 GiB at N=1.5 M / E=7.54 M / H=128 / L=8);
#   "lean"   P and t are NOT kept: the backward of a layer first re-runs the two kernels that produced them
#            (node projections, then the fused edge-t kernel) from h_in / e_in, which ARE kept (e_in is the previous
#            layer's e_out).  Bit-identical results (same kernels, same inputs), about 7 GiB less per layer at the
#            size above, for two extra kernels per layer (+3.3 ms of 25).
_D_ACTIVATIONS = os.environ.get("GNM_ACTIVATIONS", "saved").strip().lower()


_OPTION_NAMES = ("FUSED", "ACTIVATIONS", "CHAIN", "TN_SIDE", "TN_SIDE_CAP", "SRC_SIDE_CAP", "TN_AT", "TN_SPLIT", "TWO_SIDED", "TWO_SIDED_FWD", "WIDE_FUSED",
                 "NODE_FUSED", "LN_SWEEP")


class Options:
    """The schedule switches of a pass (FUSED, ACTIVATIONS, CHAIN, TN_SIDE, TN_SIDE_CAP, SRC_SIDE_CAP, TN_AT, TN_SPLIT, TWO_SIDED,
    TWO_SIDED_FWD, WIDE_FUSED, NODE_FUSED, LN_SWEEP; what each selects is described where its default is defined below) as ONE immutable
    object.  They choose between schedules that compute the same thing (tests and bench.py A/B them).  model_forward /
    layer_forward read the object that is current in the calling thread -- `current()`: the innermost `with options(...)` /
    `with use(opts)` of THIS thread, else the process defaults (environment, `set_default`) -- and the autograd functions of
    models.py / layers.py keep it with the saved activations, so that the backward of a forward runs under the SAME switches
    whatever thread autograd runs it on and whatever the caller has changed since."""
    __slots__ = _OPTION_NAMES

    def __init__(self, **kw):
        for k in _OPTION_NAMES:
            object.__setattr__(self, k, kw[k])

    def __setattr__(self, k, v):
        raise AttributeError("engine.Options is immutable: use replace(), options(...) or set_default(...)")

    def replace(self, **kw) -> "Options":
        bad = [k for k in kw if k not in _OPTION_NAMES]
        if bad:
            raise _lib.GnmError(f"engine.options: unknown switch {bad}; known: {_OPTION_NAMES}")
        if "ACTIVATIONS" in kw and kw["ACTIVATIONS"] not in ("saved", "lean"):
            raise _lib.GnmError(f"activation mode {kw['ACTIVATIONS']!r}: expected 'saved' or 'lean'")
        if "TN_AT" in kw and kw["TN_AT"] not in ("now", "next", "auto"):
            raise _lib.GnmError(f"TN_AT {kw['TN_AT']!r}: expected 'now', 'next' or 'auto'")
        d = {k: getattr(self, k) for k in _OPTION_NAMES}
        d.update(kw)
        return Options(**d)

    def __repr__(self):
        return "Options(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in _OPTION_NAMES) + ")"


_tls = threading.local()
_default: Optional[Options] = None          # built from the _D_* definitions at the end of this module


def current() -> Options:
    """The options of the calling thread (see Options)."""
    o = getattr(_tls, "opts", None)
    return o if o is not None else _default


@contextlib.contextmanager
def use(opts: Optional[Options]):
    """Run the body under `opts` in this thread (None: leave things as they are)."""
    if opts is None:
        yield current()
        return
    prev = getattr(_tls, "opts", None)
    _tls.opts = opts
    try:
        yield opts
    finally:
        _tls.opts = prev


@contextlib.contextmanager
def options(**kw):
    """`with engine.options(TWO_SIDED=False, CHAIN=False): ...` -- the current options of this thread with some switches
    changed, for the body; restored on exit whatever happens inside.  Thread-local: another thread never sees them."""
    with use(current().replace(**kw)) as o:
        yield o


def set_default(**kw) -> None:
    """Change the process-wide defaults (what a thread outside any options() / use() block runs under): start-up configuration."""
    global _default
    _default = _default.replace(**kw)


def _scoped(saved_arg: Optional[int] = None):
    """Give a pass an `opts=` keyword: the Options it runs under.  A backward pass without one runs under the options its
    forward left in the saved state (positional argument `saved_arg`)."""
    def deco(fn):
        @functools.wraps(fn)
        def w(*a, opts: Optional[Options] = None, **k):
            if opts is None and saved_arg is not None and len(a) > saved_arg:
                opts = getattr(a[saved_arg], "opts", None)
                scoped = getattr(_tls, "opts", None)
                if opts is not None and scoped is not None and scoped is not opts and repr(scoped) != repr(opts):
                    # `with engine.options(...)` around a backward only: the saved forward options win (the backward kernels must
                    # match the activations the forward kept) -- say so instead of silently comparing two identical paths
                    import warnings
                    warnings.warn("engine: this backward runs under the options its forward saved "
                                  f"({opts!r}); the options scoped around the backward call ({scoped!r}) are ignored -- scope the "
                                  "forward, or pass opts= explicitly", stacklevel=2)
            if opts is None:
                return fn(*a, **k)
            with use(opts):
                return fn(*a, **k)
        return w
    return deco


def __getattr__(name):          # engine.CHAIN, engine.ACTIVATIONS, ...: the current value (read-only; PEP 562)
    if name in _OPTION_NAMES:
        return getattr(current(), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def set_activation_mode(mode: str) -> None:
    """Process-wide default of Options.ACTIVATIONS ('saved' / 'lean')."""
    set_default(ACTIVATIONS=mode)



def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device_of(pick):
    """Run the wrapped entry point with the HIP device of `pick(*args)` current.  HIP launches go to the CURRENT
    device and _stream() is the current device's stream; the reference's own pattern is model.to('cuda:3') with
    no set_device (hyperparameters.py:25), so without this the kernels would be queued on device 0 with
    device-3 pointers."""
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **k):
            t = pick(*a, **k)
            dev = t.device if torch.is_tensor(t) else torch.device(t)
            if dev.type != "cuda":
                raise _lib.GnmError("gnnome_assembly_amd: tensors must be on a HIP device (no CPU fallback)")
            if dev.index is None or dev.index == torch.cuda.current_device():
                return fn(*a, **k)
            with torch.cuda.device(dev):
                return fn(*a, **k)
        return wrapper
    return deco


# Optional per-op timing (bench.py / profiling only): when `_prof` is a list every C-ABI call is
# bracketed by HIP events recorded on the stream the kernels are launched on.
_prof = None


def profile_ops(enable: bool):
    """Start (True) or stop (False) per-op event timing; stop returns {op: (calls, total_ms)}."""
    global _prof
    if enable:
        _prof = []
        return None
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in rec or []:
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + 1, t + e0.elapsed_time(e1))
    return out


def _call(name: str, *args, tag: str = None):
    fn = getattr(_lib.load(), name)
    if _prof is None:
        rc = fn(*args)
    else:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        _prof.append((tag or name, e0, e1))
    _lib.check(rc, name)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GnmError("gnnome_assembly_amd: tensors must be on a HIP device (no CPU fallback)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.GnmError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


class _Scratch:
    """Per-device scratch buffers (torch-owned; the library never allocates)."""

    def __init__(self, device):
        lib = _lib.load()
        self.device = device
        self.max_blocks = lib.gnm_max_partial_blocks()
        # (max_blocks + 1) rows: the extra row is the reduction scratch of gnm_bn_*finalize (gnm.h)
        self.partials = torch.empty((self.max_blocks + 1) * 2 * 256, dtype=torch.float64, device=device)
        self._ws = torch.empty(1 << 20, dtype=torch.uint8, device=device)

    def ws(self, nbytes: int) -> torch.Tensor:
        if self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self._ws


_scratch: Dict[tuple, _Scratch] = {}


def scratch(device, key: str = "main") -> _Scratch:
    device = torch.device(device)
    if (device, key) not in _scratch:
        _scratch[(device, key)] = _Scratch(device)
    return _scratch[(device, key)]


_side_streams: Dict[torch.device, "torch.cuda.Stream"] = {}


def _side_stream(device):
    device = torch.device(device)
    if device not in _side_streams:
        _side_streams[device] = torch.cuda.Stream(device=device)
    return _side_streams[device]


_side_held: Dict[torch.device, list] = {}       # what the side stream of a device is reading (kept alive until the main stream has waited)


def side_begin(device):
    """The side stream of `device`, ready for one more launch: the main stream first waits for what ran there before (a layer ago:
    free) -- which also makes the held tensors safe to drop -- and the side stream for the main stream's work so far."""
    device = torch.device(device)
    side = _side_stream(device)
    main = torch.cuda.current_stream(device)
    main.wait_stream(side)
    _side_held.setdefault(device, []).clear()
    side.wait_stream(main)
    return side


def tn_on_side(device, A, B, C_, out, first: bool) -> None:
    """gemm_tn_colsum(A, B, C_, out) on the side stream of `device` (a weight gradient: no consumer before the optimizer step), its
    operands held until the main stream has waited for it.  first: the first deferred launch since the main stream last waited
    (side_begin); a further one only makes the side stream wait for the main stream's work so far."""
    device = torch.device(device)
    if first:
        side = side_begin(device)
    else:
        side = _side_stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        gemm_tn_colsum(A, B, C_, out, sc_key="tn")
    _side_held[device].extend((A, B, C_, out))


def side_drain(device) -> None:
    """The main stream waits for the side stream's work (layer_backward(..., defer_tn=True)); the held tensors are released."""
    device = torch.device(device)
    if device in _side_streams:
        torch.cuda.current_stream(device).wait_stream(_side_streams[device])
    _side_held.get(device, []).clear()


# ---------------------------------------------------------------------------------------
# thin wrappers
# ---------------------------------------------------------------------------------------

@on_device_of(lambda mode, A, *a, **k: A)
def gemm(mode: int, A: torch.Tensor, B: torch.Tensor, C_: torch.Tensor, bias=None, resid=None, relu=False):
    """C = op(A) op(B) (+bias +resid, relu).  A, B, C, resid are 2-D views with unit inner
    stride; shapes follow include/gnm.h (NT: A[M,K] B[N,K]; NN: A[M,K] B[K,N]; TN: A[K,M] B[K,N])."""
    lib = _lib.load()
    _chk_dev(A, B, C_, bias, resid)
    for t in (A, B, C_, resid):
        if t is not None and (t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32):
            raise _lib.GnmError("gemm: operands must be 2-D float32 with unit inner stride")
    if mode == NT:
        M, K = A.shape
        N, K2 = B.shape
    elif mode == NN:
        M, K = A.shape
        K2, N = B.shape
    else:
        K, M = A.shape
        K2, N = B.shape
    if K != K2 or tuple(C_.shape) != (M, N):
        raise _lib.GnmError(f"gemm: shape mismatch mode={mode} A={tuple(A.shape)} B={tuple(B.shape)} C={tuple(C_.shape)}")
    need = lib.gnm_gemm_f32_workspace_bytes(mode, M, N, K)
    ws = scratch(A.device).ws(need) if need else None
    _call("gnm_gemm_f32", mode, M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C_),
                                C_.stride(0), _ptr(bias), _ptr(resid),
                                resid.stride(0) if resid is not None else 0, int(bool(relu)),
                                _ptr(ws), need, _stream(),
          tag=f"gemm_{('NT', 'NN', 'TN')[mode]}[{M}x{N}x{K}]" if _prof is not None else None)
    return C_


@on_device_of(lambda A, *a, **k: A)
def gemm_tn_colsum(A: torch.Tensor, B: torch.Tensor, C_: torch.Tensor, out: Optional[torch.Tensor] = None,
                   sc_key: str = "main") -> torch.Tensor:
    """C = A^T B and the column sums of A (weight and bias gradient of a Linear: A = grad of its output [rows, out],
    B = its input [rows, in]); returns the column sums.  sc_key: which scratch workspace the call uses (a launch on the side
    stream must not share the main stream's)."""
    lib = _lib.load()
    _chk_dev(A, B, C_, out)
    for t in (A, B, C_):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
            raise _lib.GnmError("gemm_tn_colsum: operands must be 2-D float32 with unit inner stride")
    K, M = A.shape
    K2, N = B.shape
    if K != K2 or tuple(C_.shape) != (M, N):
        raise _lib.GnmError(f"gemm_tn_colsum: shape mismatch A={tuple(A.shape)} B={tuple(B.shape)} C={tuple(C_.shape)}")
    if out is None:
        out = torch.empty(M, dtype=torch.float32, device=A.device)
    need = lib.gnm_gemm_tn_colsum_workspace_bytes(M, N, K)
    ws = scratch(A.device, sc_key).ws(need) if need else None
    _call("gnm_gemm_tn_colsum", M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C_), C_.stride(0), _ptr(out),
          _ptr(ws), need, _stream(), tag=f"gemm_TN+colsum[{M}x{N}x{K}]" if _prof is not None else None)
    return out


@on_device_of(lambda X, *a, **k: X)
def colsum(X: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _chk_dev(X)
    M, W = X.shape
    if out is None:
        out = torch.empty(W, dtype=torch.float32, device=X.device)
    need = lib.gnm_colsum_workspace_bytes(M, W)
    ws = scratch(X.device).ws(need)
    _call("gnm_colsum_f32", M, W, _ptr(X), X.stride(0), _ptr(out), _ptr(ws), need, _stream())
    return out


def bn_finalize(partials, nblk, count, H, gamma, beta):
    lib = _lib.load()
    stat = torch.empty(4, H, dtype=torch.float32, device=gamma.device)
    _call("gnm_bn_finalize", _ptr(partials), nblk, count, H, _ptr(gamma), _ptr(beta), EPS_BN,
                                   _ptr(stat), _stream())
    return stat


def bn_bwd_finalize(partials, nblk, count, H, device, gg=None, gb=None):
    lib = _lib.load()
    bstat = torch.empty(2, H, dtype=torch.float32, device=device)
    gg = torch.empty(H, dtype=torch.float32, device=device) if gg is None else gg
    gb = torch.empty(H, dtype=torch.float32, device=device) if gb is None else gb
    _call("gnm_bn_bwd_finalize", _ptr(partials), nblk, count, H, _ptr(bstat), _ptr(gg), _ptr(gb),
                                       _stream())
    return bstat, gg, gb


def node_rows_in(idx, x: torch.Tensor) -> torch.Tensor:
    """[N,W] rows in the caller's node numbering -> the graph's internal numbering (a gather through idx['nperm'];
    the tensor itself when the index keeps the caller's numbering)."""
    nperm = idx.get("nperm")
    if nperm is None:
        return x
    out = torch.empty_like(x)
    _call("gnm_gather_rows_f32", x.shape[0], x.shape[1], _ptr(x), _ptr(nperm), _ptr(out), _stream())
    return out


def node_rows_out(idx, x: torch.Tensor) -> torch.Tensor:
    """The inverse of node_rows_in: internal numbering -> the caller's."""
    nrank = idx.get("nrank")
    if nrank is None:
        return x
    out = torch.empty_like(x)
    _call("gnm_gather_rows_f32", x.shape[0], x.shape[1], _ptr(x), _ptr(nrank), _ptr(out), _stream())
    return out


# ---------------------------------------------------------------------------------------
# one GatedGCN layer
# ---------------------------------------------------------------------------------------

@dataclass
class LayerParams:
    """Views of one GatedGCN_1d's parameters (gated_gcn_full.py:44-59)."""
    W5: torch.Tensor       # [5H,H] = cat(A_1,A_2,A_3,B_1,B_2).weight
    b5: torch.Tensor       # [5H]
    W3: torch.Tensor       # B_3.weight [H,H]
    b3: torch.Tensor
    gamma_e: torch.Tensor
    beta_e: torch.Tensor
    gamma_h: torch.Tensor
    beta_h: torch.Tensor


@dataclass
class LayerSaved:
    h_in: torch.Tensor = None
    e_in: torch.Tensor = None
    P: torch.Tensor = None
    t: torch.Tensor = None
    stat_e: torch.Tensor = None
    e_out: torch.Tensor = None
    hf: torch.Tensor = None
    inv_f: torch.Tensor = None
    hb: torch.Tensor = None
    inv_b: torch.Tensor = None
    z: torch.Tensor = None
    stat_h: torch.Tensor = None
    opts: "Options" = None       # the switches the forward ran under: its backward runs under the same (see Options)
    chunks: list = None          # a layer wider than 256: the saved state of its 256-column problems (see WIDE_CHUNK)
    matmul: str = None           # the matmul mode (process-wide, not an Option) the forward ran in: its backward must run in the same


def _proj_and_t(idx, N, E, H, prm, h_in, e_in, nblk):
    """P = h W5^T + b5 [N,5H] and t = e W3^T + b3 + B1h[src] + B2h[dst] [E,H] (gated_gcn_full.py:107-113,120-121);
    leaves the BatchNorm partial sums of t in the scratch buffer and their count in nblk.  H is the layer's OUTPUT
    width; h_in / e_in may be narrower or wider (in_channels != out_channels: the generic GEMM route)."""
    lib = _lib.load()
    dev = h_in.device
    sc = scratch(dev)
    st = _stream()
    P = torch.empty(N, 5 * H, dtype=torch.float32, device=dev)
    t = torch.empty(E, H, dtype=torch.float32, device=dev)
    if H == 128 and current().FUSED and h_in.shape[1] == H:
        # W-stationary fused MFMA path: projections, then t + BatchNorm partials in one pass
        need = lib.gnm_rowtile_workspace_bytes(5 * H)
        ws = sc.ws(need)
        _call("gnm_node_proj_fwd", N, H, 5 * H, _ptr(h_in), _ptr(prm.W5), _ptr(prm.b5), _ptr(P), _ptr(ws), need, st)
        _call("gnm_edge_t_fused_fwd", E, H, _ptr(e_in), _ptr(prm.W3), _ptr(prm.b3), _ptr(P), _ptr(idx["isrc"]),
              _ptr(idx["idst"]), _ptr(t), _ptr(sc.partials), C.byref(nblk), _ptr(ws), need, st)
    elif H == 256 and current().FUSED and current().WIDE_FUSED and e_in.shape[1] == H and _lib.split_mode():
        # the reference's default width: projections through the split-mode GEMM, t + BatchNorm partials in one pass
        # (a workgroup keeps one 128-column half of W3 stationary in eight waves)
        gemm(NT, h_in, prm.W5, P, bias=prm.b5)
        need = lib.gnm_rowtile_workspace_bytes(5 * H)
        ws = sc.ws(need)
        _call("gnm_edge_t_fused_fwd", E, H, _ptr(e_in), _ptr(prm.W3), _ptr(prm.b3), _ptr(P), _ptr(idx["isrc"]),
              _ptr(idx["idst"]), _ptr(t), _ptr(sc.partials), C.byref(nblk), _ptr(ws), need, st, tag="gnm_edge_t_fused_fwd[256]")
    else:
        # dense projections                                                    (:107-113)
        gemm(NT, h_in, prm.W5, P, bias=prm.b5)
        gemm(NT, e_in, prm.W3, t, bias=prm.b3)
        # t += B1h[src] + B2h[dst], BatchNorm statistics over all E edges       (:120-122)
        _call("gnm_edge_t_stats_fwd", E, H, _ptr(t), _ptr(P), _ptr(idx["isrc"]), _ptr(idx["idst"]),
              _ptr(sc.partials), C.byref(nblk), st)
    return P, t


@on_device_of(lambda idx, N, E, H, prm, h_in, *a, **k: h_in)
@_scoped()
def layer_forward(idx, N: int, E: int, H: int, prm: LayerParams, h_in, e_in, save: bool, batch_norm: bool = True,
                  residual: bool = True, plan: Optional[dict] = None, ln_width: Optional[int] = None):
    """GatedGCN_1d.forward (gated_gcn_full.py:99-157) on internal-order tensors.
    Returns (h_out, e_out, LayerSaved or None).  H = out_channels; h_in [N,Hin], e_in [E,Hin] with Hin != H only
    when residual is False (the reference drops the residual then: gated_gcn_full.py:41-42).
    plan (graph.sweep_plan(device, 2), BatchNorm, H = 128 or 256): gate + by-source aggregation as ONE two-sided sweep.
    ln_width (LayerNorm mode, a layer zero-padded to the kernel width H): the layer's real out_channels -- nn.LayerNorm
    normalises over those (gated_gcn_full.py:58-59), the dead channels are left out of the row statistics."""
    lnw = H if ln_width is None else int(ln_width)
    if residual and h_in.shape[1] != H:
        raise _lib.GnmError("layer_forward: a residual layer needs in_channels == out_channels")
    if H > WIDE_CHUNK:
        return _wide_layer_forward(idx, N, E, H, prm, h_in, e_in, save, batch_norm, residual, plan)
    nblk = C.c_int(0)
    P, t = _proj_and_t(idx, N, E, H, prm, h_in, e_in, nblk)
    return _layer_forward_tail(idx, N, E, H, prm, h_in, e_in, P, t, nblk, save, batch_norm, residual, plan, lnw)


def _layer_forward_tail(idx, N, E, H, prm, h_in, e_in, P, t, nblk, save, batch_norm, residual, plan, lnw):
    """layer_forward behind the two dense products: everything from the gate to h_out.  With BatchNorm this part is separable by
    COLUMNS (per-channel statistics, per-channel gates): a layer wider than the kernels runs it once per 256-column chunk."""
    res_e = _ptr(e_in) if residual else C.c_void_p(0)
    res_h = _ptr(h_in) if residual else C.c_void_p(0)
    lib = _lib.load()
    dev = h_in.device
    sc = scratch(dev)
    st = _stream()
    f32 = dict(dtype=torch.float32, device=dev)
    # gate, edge output, by-destination gated mean                         (:122-130)
    e_out = torch.empty(E, H, **f32)
    hf = torch.empty(N, H, **f32)
    # 256: one sweep per 128-column half (BatchNorm); LayerNorm: H = 128 (row statistics inside the sweep, no barrier)
    two_sided = plan is not None and (H in (128, 256) if batch_norm else H == 128) and current().TWO_SIDED_FWD
    inv_f = torch.empty(N, H, **f32) if (save or not two_sided) else None      # inv_f / inv_b: only the backward reads them
    if two_sided and not batch_norm:
        hb = torch.empty(N, H, **f32)
        inv_b = torch.empty(N, H, **f32) if save else None
        z = torch.empty(N, H, **f32)
        stat_e = None
        _call("gnm_ln_edge_gate2_fwd", N, E, H, _ptr(t), res_e, _ptr(prm.gamma_e), _ptr(prm.beta_e), lnw, _ptr(P), _ptr(idx["isrc"]),
              _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(plan["sinfo"]), _ptr(plan["dinfo"]), plan["nodes_per_block"], plan["nfix"],
              _ptr(plan["fix_nodes"]), _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(e_out),
              _ptr(hf), _ptr(inv_f), _ptr(hb), _ptr(inv_b), _ptr(z), _ptr(sc.partials), C.byref(nblk), st)
    elif two_sided:
        hb = torch.empty(N, H, **f32)
        inv_b = torch.empty(N, H, **f32) if save else None
        z = torch.empty(N, H, **f32)
        stat_e = bn_finalize(sc.partials, nblk.value, E, H, prm.gamma_e, prm.beta_e)
        _call("gnm_edge_gate2_fwd", N, E, H, _ptr(t), res_e, _ptr(stat_e), _ptr(P), _ptr(idx["isrc"]), _ptr(idx["idst"]),
              _ptr(idx["in_ptr"]), _ptr(plan["sinfo"]), _ptr(plan["dinfo"]), plan["nodes_per_block"], plan["nfix"],
              _ptr(plan["fix_nodes"]), _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(e_out),
              _ptr(hf), _ptr(inv_f), _ptr(hb), _ptr(inv_b), _ptr(z), _ptr(sc.partials), C.byref(nblk), st)
    elif batch_norm:
        stat_e = bn_finalize(sc.partials, nblk.value, E, H, prm.gamma_e, prm.beta_e)
        _call("gnm_edge_gate_fwd", N, E, H, _ptr(t), res_e, _ptr(stat_e), _ptr(P), _ptr(idx["isrc"]),
              _ptr(idx["in_ptr"]), _ptr(e_out), _ptr(hf), _ptr(inv_f), st)
    else:       # LayerNorm: row statistics inside the kernel, no barrier
        stat_e = None
        _call("gnm_ln_edge_gate_fwd", N, E, H, _ptr(t), res_e, _ptr(prm.gamma_e), _ptr(prm.beta_e), _ptr(P),
              _ptr(idx["isrc"]), _ptr(idx["in_ptr"]), _ptr(e_out), _ptr(hf), _ptr(inv_f), lnw, st)
    # by-source gated mean on the same gate, z, BatchNorm statistics over N (:133-147)
    if not two_sided:
        hb = torch.empty(N, H, **f32)
        inv_b = torch.empty(N, H, **f32)
        z = torch.empty(N, H, **f32)
        _call("gnm_node_agg_src_fwd", N, E, H, _ptr(e_out), _ptr(P), _ptr(idx["out_ptr"]),
              _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(hf), _ptr(hb),
              _ptr(inv_b), _ptr(z), _ptr(sc.partials), C.byref(nblk), st)
    h_out = torch.empty(N, H, **f32)
    if batch_norm:
        stat_h = bn_finalize(sc.partials, nblk.value, N, H, prm.gamma_h, prm.beta_h)
        _call("gnm_node_update_fwd", N, H, _ptr(z), _ptr(stat_h), res_h, _ptr(h_out), st)
    else:
        stat_h = None
        _call("gnm_ln_node_update_fwd", N, H, _ptr(z), _ptr(prm.gamma_h), _ptr(prm.beta_h), res_h, _ptr(h_out), lnw, st)
    saved = None
    if save:
        lean = current().ACTIVATIONS == "lean"
        saved = LayerSaved(opts=current(), h_in=h_in, e_in=e_in, P=None if lean else P, t=None if lean else t, stat_e=stat_e, e_out=e_out,
                           hf=hf, inv_f=inv_f, hb=hb, inv_b=inv_b, z=z, stat_h=stat_h, matmul=_lib.get_matmul_mode())
    return h_out, e_out, saved


# A layer wider than the widest kernel instantiation (gated_gcn_full.py:44-50 takes any nn.Linear width): the two dense products
# run at full width on the generic GEMM route, everything between them -- BatchNorm statistics, gate, both aggregations, node
# update, and the duals of all of it -- is separable by columns and runs once per WIDE_CHUNK-column problem on contiguous copies
# of the chunk (strided copies in and out: slow by construction, but a legal width no longer raises).  BatchNorm only: LayerNorm's
# row statistics span the chunks.
WIDE_CHUNK = 256


def _cols(x: torch.Tensor, c0: int, w: int, blocks: int = 1) -> torch.Tensor:
    """Columns c0 .. c0+w of each of the `blocks` equal blocks of x's rows, as a contiguous [rows, blocks*w] tensor."""
    R, Hb = x.shape[0], x.shape[1] // blocks
    return x.view(R, blocks, Hb)[:, :, c0:c0 + w].reshape(R, blocks * w).contiguous()


def _put_cols(dst: torch.Tensor, src: torch.Tensor, c0: int, w: int, blocks: int = 1) -> None:
    R, Hb = dst.shape[0], dst.shape[1] // blocks
    dst.view(R, blocks, Hb)[:, :, c0:c0 + w] = src.view(R, blocks, w)


def _chunk_params(prm: LayerParams, c0: int, w: int) -> LayerParams:
    """The per-channel parameters of one column chunk (the weights are not read behind the dense products)."""
    sl = slice(c0, c0 + w)
    return LayerParams(W5=None, b5=None, W3=None, b3=None, gamma_e=prm.gamma_e[sl], beta_e=prm.beta_e[sl],
                       gamma_h=prm.gamma_h[sl], beta_h=prm.beta_h[sl])


def _wide_layer_forward(idx, N, E, H, prm, h_in, e_in, save, batch_norm, residual, plan):
    if not batch_norm:
        raise NotImplementedError(f"GatedGCN_1d with LayerNorm at width {H}: the LayerNorm kernels hold a row in one wavefront "
                                  f"(widths up to {WIDE_CHUNK}); BatchNorm layers run at any width")
    if H % WIDE_CHUNK:
        raise _lib.GnmError(f"layer_forward: a wide layer runs zero-padded to a multiple of {WIDE_CHUNK} (layers.padded_width), got {H}")
    dev = h_in.device
    sc = scratch(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    P = torch.empty(N, 5 * H, **f32)
    t = torch.empty(E, H, **f32)
    gemm(NT, h_in, prm.W5, P, bias=prm.b5)                                      # (:107-112)
    gemm(NT, e_in, prm.W3, t, bias=prm.b3)                                      # (:113)
    h_out, e_out = torch.empty(N, H, **f32), torch.empty(E, H, **f32)
    chunks = []
    w = WIDE_CHUNK
    for c0 in range(0, H, w):
        Pc, tc = _cols(P, c0, w, 5), _cols(t, c0, w)
        hc = _cols(h_in, c0, w) if residual else h_in
        ec = _cols(e_in, c0, w) if residual else e_in
        nblk = C.c_int(0)
        _call("gnm_edge_t_stats_fwd", E, w, _ptr(tc), _ptr(Pc), _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(sc.partials), C.byref(nblk),
              _stream())
        ho, eo, sv = _layer_forward_tail(idx, N, E, w, _chunk_params(prm, c0, w), hc, ec, Pc, tc, nblk, save, True, residual, plan, w)
        _put_cols(h_out, ho, c0, w)
        _put_cols(e_out, eo, c0, w)
        if save:
            sv.h_in = sv.e_in = None            # the full-width inputs are kept once, below
            chunks.append(sv)
    saved = LayerSaved(opts=current(), h_in=h_in, e_in=e_in, chunks=chunks, matmul=_lib.get_matmul_mode()) if save else None
    return h_out, e_out, saved


def _same_matmul_mode(s) -> None:
    """A lean-mode backward rebuilds P and t with the forward's kernels, and the chained / two-sided schedules exist in the split
    modes only: a backward in another matmul mode than its forward would silently mix arithmetic."""
    m = getattr(s, "matmul", None)
    if m is not None and m != _lib.get_matmul_mode():
        raise _lib.GnmError(f"backward in matmul mode {_lib.get_matmul_mode()!r} of a forward that ran in {m!r}: "
                            "gnm_set_matmul_mode is process-wide, change it between steps, not inside one")


def _wide_layer_backward(idx, N, E, H, prm, s: LayerSaved, gh_out, ge, out, residual, plan):
    dev = gh_out.device
    f32 = dict(dtype=torch.float32, device=dev)
    Hin = s.h_in.shape[1]
    w = WIDE_CHUNK
    new = lambda key, *shape: out[key] if key in out else torch.empty(*shape, **f32)  # noqa: E731
    g: Dict[str, torch.Tensor] = {k: new(k, H) for k in ("gamma_e", "beta_e", "gamma_h", "beta_h")}
    gP = torch.empty(N, 5 * H, **f32)
    gt = torch.empty(E, H, **f32)
    ge_tot = ge if residual else None           # the residual path adds the incoming edge gradient: updated in place, chunk by chunk
    for ci, c0 in enumerate(range(0, H, w)):
        sl = slice(c0, c0 + w)
        sc_ = s.chunks[ci]
        outc = {k: g[k][sl] for k in g}
        gec = _cols(ge, c0, w)
        gPc, gec, bstat_e, _ = _bn_backward_mid(idx, N, E, w, _chunk_params(prm, c0, w), sc_, _cols(gh_out, c0, w), gec, outc, plan)
        gtc = torch.empty(E, w, **f32)
        _call("gnm_edge_bwd_gt", E, w, _ptr(gec), _ptr(sc_.t), _ptr(sc_.stat_e), _ptr(bstat_e), _ptr(prm.gamma_e[sl]), _ptr(gtc), _stream())
        _put_cols(gP, gPc, c0, w, 5)
        _put_cols(gt, gtc, c0, w)
        if residual:
            _put_cols(ge_tot, gec, c0, w)
        s.chunks[ci] = None
    g["W3"] = new("W3", H, Hin)
    g["b3"] = gemm_tn_colsum(gt, s.e_in, g["W3"], out.get("b3"))
    if residual:
        gemm(NN, gt, prm.W3, ge_tot, resid=ge_tot)
        ge_in = ge_tot
    else:
        ge_in = gemm(NN, gt, prm.W3, torch.empty(E, Hin, **f32))
    del gt
    g["W5"] = new("W5", 5 * H, Hin)
    g["b5"] = gemm_tn_colsum(gP, s.h_in, g["W5"], out.get("b5"))
    gh_in = torch.empty(N, Hin, **f32)
    gemm(NN, gP, prm.W5, gh_in, resid=gh_out if residual else None)
    return gh_in, ge_in, g


def _bn_backward_mid(idx, N, E, H, prm, s, gh_out, ge, out, plan):
    """The BatchNorm layer backward between the dense products (column-separable like _layer_forward_tail): BatchNorm_h backward,
    the by-destination and by-source passes.  `ge` ([E,H]) is updated in place to ge_tot = ge + gsigma sigma'.  Returns
    (gP [N,5H], ge, bstat_e, {gamma_h, beta_h, gamma_e, beta_e gradients})."""
    lib = _lib.load()
    dev = gh_out.device
    sc = scratch(dev)
    st = _stream()
    nblk = C.c_int(0)
    f32 = dict(dtype=torch.float32, device=dev)
    g: Dict[str, torch.Tensor] = {}
    gP = torch.empty(N, 5 * H, **f32)
    Q = torch.empty(N, 2 * H, **f32)     # Qf | Qb
    # BatchNorm_h backward statistics, then gz and the per-node gate-gradient factors
    _call("gnm_node_bwd_stats", N, H, _ptr(s.z), _ptr(s.stat_h), _ptr(gh_out), _ptr(sc.partials),
          C.byref(nblk), st)
    bstat_h, g["gamma_h"], g["beta_h"] = bn_bwd_finalize(sc.partials, nblk.value, N, H, dev, out.get("gamma_h"), out.get("beta_h"))
    _call("gnm_node_bwd_apply", N, H, _ptr(s.z), _ptr(s.stat_h), _ptr(bstat_h), _ptr(prm.gamma_h),
          _ptr(gh_out), _ptr(s.inv_f), _ptr(s.inv_b), _ptr(gP), _ptr(Q), st)
    if plan is not None and H in (128, 256) and current().TWO_SIDED:
        # the two-sided sweep of the chained schedule's top layer (H = 256: once per 128-column half, row pitch 256; H = 128:
        # the fp32-MFMA matmul mode, whose edge backward is not chained): by-destination AND by-source sums from one pass over
        # ge, e_out, t; then the unserved sources and the conversion through m1, m2
        UT = torch.empty(N, 2 * H, **f32)
        DT = torch.empty(N, 2 * H, **f32)
        Ud, Td = DT[:, :H], DT[:, H:]
        need_f = lib.gnm_edge_bwd_fused_workspace_bytes()
        ws = sc.ws(need_f)
        _call("gnm_edge_bwd_top", N, E, H, _ptr(ge), _ptr(s.e_out), _ptr(s.t), _ptr(s.stat_e), _ptr(s.P), _ptr(Q), _ptr(s.hf),
              _ptr(s.hb), _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(Ud), _ptr(Td),
              _ptr(sc.partials), _ptr(plan["sinfo"]), plan["nodes_per_block"], _ptr(UT), C.byref(nblk), _ptr(ws), need_f, st)
        _call("gnm_edge_bwd_src_fix", plan["nfix"], _ptr(plan["fix_nodes"]), N, E, H, _ptr(s.e_out), _ptr(s.t),
              _ptr(s.stat_e), _ptr(ge), _ptr(Q), _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]),
              _ptr(gP), _ptr(UT), st)
        bstat_e, g["gamma_e"], g["beta_e"] = bn_bwd_finalize(sc.partials, nblk.value, E, H, dev, out.get("gamma_e"), out.get("beta_e"))
        _call("gnm_node_bgrad", N, H, _ptr(s.stat_e), _ptr(bstat_e), _ptr(prm.gamma_e), _ptr(idx["in_ptr"]),
              _ptr(idx["out_ptr"]), _ptr(UT), _ptr(Ud), _ptr(Td), Ud.stride(0), _ptr(gP), st)
        del UT, DT
    else:
        # by-destination pass: ge <- ge + gsigma*sigma', gA3h, BatchNorm_e backward statistics
        Ud = torch.empty(N, H, **f32)
        Td = torch.empty(N, H, **f32)
        _call("gnm_edge_bwd_dst", N, E, H, _ptr(s.e_out), _ptr(s.t), _ptr(s.stat_e), _ptr(ge), _ptr(s.P),
              _ptr(Q), _ptr(s.hf), _ptr(s.hb), _ptr(idx["isrc"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(Ud), _ptr(Td),
              _ptr(sc.partials), C.byref(nblk), st)
        bstat_e, g["gamma_e"], g["beta_e"] = bn_bwd_finalize(sc.partials, nblk.value, E, H, dev, out.get("gamma_e"), out.get("beta_e"))
        # by-source pass: gA2h, gB1h, gB2h
        _call("gnm_edge_bwd_src", N, E, H, _ptr(s.e_out), _ptr(s.t), _ptr(s.stat_e), _ptr(bstat_e),
              _ptr(prm.gamma_e), _ptr(ge), _ptr(Q), _ptr(idx["in_ptr"]), _ptr(idx["out_ptr"]),
              _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(Ud), _ptr(Td), _ptr(gP), 0, st)
    del Ud, Td, Q
    return gP, ge, bstat_e, g


@on_device_of(lambda idx, N, E, H, prm, s, gh_out, *a, **k: gh_out)
@_scoped(5)
def layer_backward(idx, N: int, E: int, H: int, prm: LayerParams, s: LayerSaved, gh_out, ge, batch_norm: bool = True,
                   out: Optional[Dict[str, torch.Tensor]] = None, residual: bool = True, plan: Optional[dict] = None,
                   ln_width: Optional[int] = None, defer_tn: bool = False):
    """Backward of layer_forward.  `ge` ([E,H], internal order) holds d loss / d e_out on entry
    and is OVERWRITTEN with d loss / d e_in (residual layers; without the residual the returned ge is a fresh [E,Hin]
    tensor).  Returns (gh_in, ge, grads dict).  `out` (optional) names the tensors
    the parameter gradients are written INTO (keys W5 b5 W3 b3 gamma_e beta_e gamma_h beta_h; contiguous blocks,
    e.g. views of a flat gradient buffer) instead of fresh allocations.
    defer_tn (H = 128 fused route): the node-projection weight gradient may run on the side stream; the CALLER then calls
    side_drain(device) before anything reads W5 / b5 gradients (model_backward does, after the last layer)."""
    out = out or {}
    Hin = s.h_in.shape[1]
    lnw = H if ln_width is None else int(ln_width)
    fused = H == 128 and current().FUSED and residual          # the fused backward kernels have the residual adds built in
    # the two weight-gradient GEMMs of a 256-wide layer on the side stream (model_backward drains): see tn_on_side
    defer_side = (defer_tn and H == 256 and batch_norm and current().TN_SIDE and _prof is None and current().ACTIVATIONS != "lean"
                  and _lib.split_mode())
    deferred_first = True
    new = lambda key, *shape: out[key] if key in out else torch.empty(*shape, dtype=torch.float32, device=gh_out.device)  # noqa: E731
    lib = _lib.load()
    dev = gh_out.device
    sc = scratch(dev)
    st = _stream()
    nblk = C.c_int(0)
    f32 = dict(dtype=torch.float32, device=dev)
    g: Dict[str, torch.Tensor] = {}
    _same_matmul_mode(s)
    if s.chunks is not None:
        return _wide_layer_backward(idx, N, E, H, prm, s, gh_out, ge, out, residual, plan)
    if s.P is None or s.t is None:      # "lean" activations: rebuild P and t with the kernels that made them
        s.P, s.t = _proj_and_t(idx, N, E, H, prm, s.h_in, s.e_in, C.c_int(0))
    g["W3"] = new("W3", H, Hin)
    if not batch_norm:
        gP = torch.empty(N, 5 * H, **f32)
        Q = torch.empty(N, 4 * H, **f32)     # LayerNorm mode: Qf | Qb | Rf | Rb
        # ---- LayerNorm mode: no global barriers, gt is produced by the by-destination pass ----
        _call("gnm_ln_node_bwd", N, H, _ptr(s.z), _ptr(prm.gamma_h), _ptr(prm.beta_h), _ptr(gh_out), _ptr(s.hf),
              _ptr(s.inv_f), _ptr(s.hb), _ptr(s.inv_b), _ptr(gP), _ptr(Q), _ptr(sc.partials), C.byref(nblk), lnw, st)
        _, g["gamma_h"], g["beta_h"] = bn_bwd_finalize(sc.partials, nblk.value, N, H, dev, out.get("gamma_h"), out.get("beta_h"))
        gt = torch.empty(E, H, **f32)
        if plan is not None and H == 128 and current().TWO_SIDED and current().LN_SWEEP:
            # round 6: by-destination AND by-source sums from ONE two-sided sweep (the LayerNorm form of the top sweep: gt is complete
            # inside a row, so gB1h / gB2h come out of the sweep directly), then the plan's unserved sources by gathers
            need_f = lib.gnm_edge_bwd_fused_workspace_bytes()
            ws = sc.ws(need_f)
            _call("gnm_ln_edge_bwd_top", N, E, H, _ptr(ge), _ptr(s.e_out), _ptr(s.t), _ptr(prm.gamma_e), _ptr(prm.beta_e), lnw,
                  _ptr(s.P), _ptr(Q), _ptr(s.hf), _ptr(s.hb), _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(gt),
                  _ptr(sc.partials), _ptr(plan["sinfo"]), plan["nodes_per_block"], C.byref(nblk), _ptr(ws), need_f, st)
            _call("gnm_ln_edge_bwd_src_fix", plan["nfix"], _ptr(plan["fix_nodes"]), N, E, H, _ptr(s.e_out), _ptr(gt), _ptr(Q),
                  _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(gP), st)
            _, g["gamma_e"], g["beta_e"] = bn_bwd_finalize(sc.partials, nblk.value, E, H, dev, out.get("gamma_e"), out.get("beta_e"))
        else:
            _call("gnm_ln_edge_bwd_dst", N, E, H, _ptr(s.e_out), _ptr(s.t), _ptr(prm.gamma_e), _ptr(prm.beta_e),
                  _ptr(ge), _ptr(s.P), _ptr(Q), _ptr(idx["isrc"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(gt),
                  _ptr(sc.partials), C.byref(nblk), lnw, st)
            _, g["gamma_e"], g["beta_e"] = bn_bwd_finalize(sc.partials, nblk.value, E, H, dev, out.get("gamma_e"), out.get("beta_e"))
            _call("gnm_ln_edge_bwd_src", N, E, H, _ptr(s.e_out), _ptr(gt), _ptr(Q), _ptr(idx["out_ptr"]),
                  _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(gP), st)
        del Q
        if fused and Hin == H and _lib.split_mode():
            # round 6: gW3 = gt^T e_in, gb3 = sum gt and ge_in = ge + gt W3 from ONE pass over gt, ge, e_in (the fused edge backward
            # with gt given) instead of gemm_tn_colsum + gemm NN with the residual add
            g["b3"] = new("b3", H)
            need = lib.gnm_edge_bwd_fused_workspace_bytes()
            ws = sc.ws(need)
            _call("gnm_edge_bwd_fused_gt", E, H, _ptr(ge), _ptr(ge), _ptr(gt), _ptr(s.e_in), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]),
                  _ptr(sc.partials), _ptr(ws), need, st)
        else:
            g["b3"] = gemm_tn_colsum(gt, s.e_in, g["W3"], out.get("b3"))
            if residual:
                gemm(NN, gt, prm.W3, ge, resid=ge)
            else:
                ge = gemm(NN, gt, prm.W3, torch.empty(E, Hin, **f32))
        del gt
    else:
        gP, ge, bstat_e, gm = _bn_backward_mid(idx, N, E, H, prm, s, gh_out, ge, out, plan)
        g.update(gm)
        # gt, B_3 gradients, ge_in = ge_tot + gt W3
        if fused:
            g["b3"] = new("b3", H)
            need = lib.gnm_edge_bwd_fused_workspace_bytes()
            ws = sc.ws(need)
            _call("gnm_edge_bwd_fused", E, H, _ptr(ge), _ptr(ge), _ptr(s.t), _ptr(s.e_in), _ptr(s.stat_e), _ptr(bstat_e),
                  _ptr(prm.gamma_e), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]), _ptr(sc.partials), _ptr(ws), need, st)
        elif H == 256 and current().FUSED and current().WIDE_FUSED and residual and Hin == H and _lib.split_mode():
            # the reference's default width: gt and ge_in = ge_tot + gt W3 from one pass, gt kept for the weight-gradient GEMM
            gt = torch.empty(E, H, **f32)
            ge_in = torch.empty(E, H, **f32)
            need = lib.gnm_rowtile_workspace_bytes(5 * H)
            ws = sc.ws(need)
            _call("gnm_edge_bwd_gt_nn", E, H, _ptr(ge), _ptr(s.t), _ptr(s.stat_e), _ptr(bstat_e), _ptr(prm.gamma_e),
                  _ptr(prm.W3), _ptr(gt), _ptr(ge_in), _ptr(ws), need, st)
            ge = ge_in
            if defer_side:          # round 6: beside the next layer's HBM-bound sweep instead of in front of it
                g["b3"] = new("b3", H)
                tn_on_side(dev, gt, s.e_in, g["W3"], g["b3"], first=True)
                deferred_first = False
            else:
                g["b3"] = gemm_tn_colsum(gt, s.e_in, g["W3"], out.get("b3"))
            del gt
        else:
            gt = torch.empty(E, H, **f32)
            _call("gnm_edge_bwd_gt", E, H, _ptr(ge), _ptr(s.t), _ptr(s.stat_e), _ptr(bstat_e),
                  _ptr(prm.gamma_e), _ptr(gt), st)
            g["b3"] = gemm_tn_colsum(gt, s.e_in, g["W3"], out.get("b3"))
            if residual:
                gemm(NN, gt, prm.W3, ge, resid=ge)
            else:
                ge = gemm(NN, gt, prm.W3, torch.empty(E, Hin, **f32))
            del gt
    # node projections backward
    g["W5"] = new("W5", 5 * H, Hin)
    gh_in = torch.empty(N, Hin, **f32)
    if fused:
        g["b5"] = new("b5", 5 * H)
        need = lib.gnm_node_proj_bwd_workspace_bytes(5 * H)
        ws = sc.ws(need)
        _call("gnm_node_proj_bwd_nn", N, H, 5 * H, _ptr(gP), _ptr(prm.W5), _ptr(gh_out), _ptr(gh_in), _ptr(ws), need, st)
        if defer_tn and current().TN_SIDE and _prof is None and current().ACTIVATIONS != "lean":
            # the weight gradient has no consumer before the optimizer step: on the side stream, beside the next layer's HBM-bound
            # node / by-destination passes (what the chained schedule does with its deferred kernel); the caller drains (side_drain)
            side = side_begin(dev)
            sc3 = scratch(dev, "tn")
            ws3 = sc3.ws(need)
            _lib.check(lib.gnm_node_proj_bwd_tn(N, H, 5 * H, _ptr(gP), _ptr(s.h_in), _ptr(g["W5"]), _ptr(g["b5"]), _ptr(sc3.partials),
                                                _ptr(ws3), need, current().TN_SIDE_CAP, C.c_void_p(side.cuda_stream)), "gnm_node_proj_bwd_tn")
            _side_held[torch.device(dev)].extend((gP, s.h_in, g["W5"], g["b5"]))
        else:
            _call("gnm_node_proj_bwd_tn", N, H, 5 * H, _ptr(gP), _ptr(s.h_in), _ptr(g["W5"]), _ptr(g["b5"]),
                  _ptr(sc.partials), _ptr(ws), need, 0, st)
    else:
        gemm(NN, gP, prm.W5, gh_in, resid=gh_out if residual else None)
        if defer_side:
            g["b5"] = new("b5", 5 * H)
            tn_on_side(dev, gP, s.h_in, g["W5"], g["b5"], first=deferred_first)
        else:
            g["b5"] = gemm_tn_colsum(gP, s.h_in, g["W5"], out.get("b5"))
    return gh_in, ge, g


# The chained backward schedule (H = 128, BatchNorm, the split matmul modes): the fused edge backward of layer i runs in
# ONE kernel with the by-destination pass of layer i-1 (gnm_edge_bwd_chain: 5 [E,H] streams instead of 4 + 4, the
# matrix-core work under the gather arithmetic).  GNM_CHAIN=0 / engine.options(CHAIN=False) goes back to layer_backward.
_D_CHAIN = os.environ.get("GNM_CHAIN", "1") != "0"
# Inside the chained schedule: layer i's node-projection weight gradient (gW5 = gP^T h_in, matrix-core bound, no consumer
# before the optimizer step) is launched on a side stream BESIDE layer i-1's by-source pass (HBM bound), which is capped
# at four workgroups per CU (4 x 80 registers per SIMD lane) so that the weight-gradient workgroups (168 registers) find
# room on every CU.  Measured on one box: 187.9 -> 183.8 ms/step; with five by-source workgroups the two kernels no
# longer co-reside and nothing is gained.  The caps travel as ARGUMENTS of the two launches (max_blocks_per_cu; 0 = none):
# no process-wide state is touched (round 2 flipped gnm_set_occupancy_cap between the launches, and the weight-gradient
# kernel never honoured it: TN_SIDE_CAP = 0 is what was measured).  GNM_TN_SIDE=0 keeps everything on one stream (the
# per-op timing mode always does).
_D_TN_SIDE = os.environ.get("GNM_TN_SIDE", "1") != "0"
_D_TN_SIDE_CAP = int(os.environ.get("GNM_TN_CAP", "0"))
# when the deferred weight-gradient kernel of layer i is launched: "next" = at the head of layer i-1's iteration (beside its
# by-source pass / conversion), "now" = right after layer i's own by-source pass or conversion (beside nn(i), node(i-1))
# "auto" (default): by graph size -- "now" from TN_AT_NOW_NODES nodes on, "next" below.  At the metric's graph (N = 1.5 M) "now" wins
# (161.4 vs 163.2 ms/step bf16x3, 153.6-154.2 vs 154.8-155.4 f16x2: profiles/r05_ab_tn_at.txt, r05_ab_schedule_f16x2.txt); on small graphs the
# HBM-bound window beside nn(i) / node(i-1) is too short for the weight-gradient kernel and "next" wins: N = 750 k 76.8-77.7 vs 77.4-79.0 ms,
# N = 220 k (the true chr19 size) 24.6-24.7 vs 24.8-24.9, the mini-batch epoch (sub-graphs of N = 150 k) 41.5-42.1 vs 40.5-41.3 M edges/s
# (profiles/r05_ab_tn_at_sizes.txt)
_D_TN_AT = os.environ.get("GNM_TN_AT", "auto")
TN_AT_NOW_NODES = int(os.environ.get("GNM_TN_AT_NOW_NODES", "1100000"))


def tn_at(N: int) -> str:
    """The TN_AT switch of the current options resolved for a graph of N nodes ("auto": by size, see TN_AT_NOW_NODES)."""
    v = current().TN_AT
    return ("now" if N >= TN_AT_NOW_NODES else "next") if v == "auto" else v
# "next" only: the deferred kernel in TWO launches sized to the two HBM-bound windows of an iteration -- the gB1h | gB2h column groups
# beside this layer's conversion (node_bgrad), the gA1h | gA2h | gA3h groups AFTER this layer's nn (both matrix bound: side by side
# they only take turns) beside the next layer's BatchNorm_h backward.  GNM_TN_SPLIT=0: one launch at the head of the iteration.
_D_TN_SPLIT = os.environ.get("GNM_TN_SPLIT", "1") != "0"
_D_SRC_SIDE_CAP = int(os.environ.get("GNM_SRC_CAP", "4"))


# The chained kernel as a TWO-SIDED sweep (gnm_edge_bwd_chain_src): layer i-1's by-source sums (gA2h, Us, Ts) come out of the
# same pass through the graph's sweep plan (graph.sweep_plan), the separate by-source pass -- three [E,H] streams re-read per
# layer -- shrinks to a gather over the few per cent of the nodes the plan does not serve plus an [N,H]-sized conversion
# once the BatchNorm-backward means are known.  GNM_TWO_SIDED=0 / engine.options(TWO_SIDED=False) keeps the separate pass.
_D_TWO_SIDED = os.environ.get("GNM_TWO_SIDED", "1") != "0"
# the forward twin (gnm_edge_gate2_fwd): gate + by-destination AND by-source aggregation in one sweep; GNM_TWO_SIDED_FWD=0
# keeps edge_gate_fwd + node_agg_src_fwd
_D_TWO_SIDED_FWD = os.environ.get("GNM_TWO_SIDED_FWD", "1") != "0"
# H = 256 (the reference's default width): the fused forward kernel for t (gnm_edge_t_fused_fwd at H = 256);
# GNM_WIDE_FUSED=0 keeps gemm NT + edge_t_stats_fwd
_D_WIDE_FUSED = os.environ.get("GNM_WIDE_FUSED", "1") != "0"


# Round 5, the node side.  (A pre-split image of h -- the kernel that produces h also writes its three bf16 terms, the projections
# and their weight gradient copy them instead of splitting the rows in every workgroup class -- was built, measured at +1.4 ms per
# step, profiles/r05_ab_node_side.txt, and removed again when the f16x2 mode made its bf16x3 layout the wrong one: DESIGN.md 3g.)
# NODE_FUSED (chained schedule with a sweep plan): no gnm_node_bgrad / gnm_node_bwd_stats launches -- the conversion of the raw
# by-source / by-destination sums runs in the operand load of the weight-gradient kernel of those two column groups
# (gnm_tn128_bgrad, which also writes them for the projection backward behind it), the BatchNorm_h backward sums of the layer
# below in the epilogue of the projection backward (gnm_node_proj_bwd_nn_stats).  GNM_NODE_FUSED=0: the round-4 schedule.
_D_NODE_FUSED = os.environ.get("GNM_NODE_FUSED", "1") != "0"
# Round 6, LayerNorm models at H = 128 with a sweep plan: the backward's by-destination and by-source passes as ONE two-sided sweep
# (gnm_ln_edge_bwd_top + gnm_ln_edge_bwd_src_fix).  GNM_LN_SWEEP=0: gnm_ln_edge_bwd_dst + gnm_ln_edge_bwd_src.
_D_LN_SWEEP = os.environ.get("GNM_LN_SWEEP", "1") != "0"


def tn128(N: int, A: torch.Tensor, lda: int, ncg: int, h: torch.Tensor, W, b, partials, ws, need, stream=None, tag: str = None):
    """gW[cg] = A[:, cg]^T h, gb[cg] = sum A[:, cg] (gnm_tn128).  `stream`: a torch side stream (the call is then not profiled);
    None = the current stream."""
    if stream is None:
        _call("gnm_tn128", N, _ptr(A), lda, ncg, _ptr(h), _ptr(W), _ptr(b), _ptr(partials), _ptr(ws), need, _stream(), tag=tag)
    else:
        _lib.check(_lib.load().gnm_tn128(N, _ptr(A), lda, ncg, _ptr(h), _ptr(W), _ptr(b), _ptr(partials), _ptr(ws), need,
                                         C.c_void_p(stream.cuda_stream)), "gnm_tn128")


# workgroups per CU of the two-sided forward sweep (the partition its plan is built for); 1 only together with
# GNM_VARIANTS=gate2_wg=1: the occupancy experiment of profiles/r06_ab_gate2_occupancy.txt
GATE2_WG = int(os.environ.get("GNM_GATE2_WG", "2"))


def sweep_width(H: int, batch_norm: bool = True) -> bool:
    """Does a layer of (kernel) width H run the two-sided sweeps?  128; with BatchNorm also 256 (one sweep per 128-column half) and
    the wide layers (256-column chunks)."""
    return H == 128 or (batch_norm and (H == 256 or (H > WIDE_CHUNK and H % WIDE_CHUNK == 0)))


def chain_eligible(H: int, batch_norm: bool) -> bool:
    return current().CHAIN and current().FUSED and H == 128 and batch_norm and _lib.split_mode()


def layers_backward_chained(idx, N: int, E: int, H: int, P: Dict[str, torch.Tensor], L: int, saved: List[LayerSaved],
                            gh, ge, outs: List[Optional[Dict[str, torch.Tensor]]], plan: Optional[dict] = None):
    """Backward of the L-layer stack (layers L-1 .. 0), same arithmetic as L x layer_backward, other schedule:
        node(L-1), dst(L-1);   then for i = L-1 .. 0:   finalize_e(i), src(i), proj(i),
                                                         i > 0:  node(i-1), CHAIN[fused(i) + dst(i-1)]
                                                         i = 0:  fused(0)
    `ge` is updated in place through all layers.  Returns (gh_in of layer 0, ge_in of layer 0, [grads dict per layer]);
    saved[i] is released as soon as layer i is done.  outs[i]: write-into targets as in layer_backward (or None).
    With `plan` (graph.sweep_plan) the chained kernel is the two-sided sweep: src(i) for i < L-1 becomes
    fix(i) [right after CHAIN(i+1, i)] + bgrad(i) [after finalize_e(i)]."""
    lib = _lib.load()
    dev = gh.device
    sc, sc2 = scratch(dev), scratch(dev, "side")
    st = _stream()
    f32 = dict(dtype=torch.float32, device=dev)
    grads: List[Dict[str, torch.Tensor]] = [dict() for _ in range(L)]
    prms = [None] * L

    def tgt(i, key, *shape):
        o = outs[i] or {}
        return o[key] if key in o else torch.empty(*shape, **f32)

    def ensure(i):
        if prms[i] is None:
            prms[i] = layer_params(P, i)
        s = saved[i]
        if s.P is None or s.t is None:      # "lean" activations
            s.P, s.t = _proj_and_t(idx, N, E, H, prms[i], s.h_in, s.e_in, C.c_int(0))
        return prms[i], s

    def node(i, gh_out, nblk_h=None):
        """BatchNorm_h backward of layer i -> gP[:, 0:H] = gz, Q = Qf | Qb.  nblk_h: the (sum gw, sum gw zhat) partials are in
        the scratch buffer already (the projection backward of the layer above took them in its epilogue: NODE_FUSED)."""
        prm, s = ensure(i)
        o = outs[i] or {}
        gP = torch.empty(N, 5 * H, **f32)
        Q = torch.empty(N, 2 * H, **f32)
        if nblk_h is None:
            nblk_h = C.c_int(0)
            _call("gnm_node_bwd_stats", N, H, _ptr(s.z), _ptr(s.stat_h), _ptr(gh_out), _ptr(sc.partials), C.byref(nblk_h), st)
        bstat_h, grads[i]["gamma_h"], grads[i]["beta_h"] = bn_bwd_finalize(sc.partials, nblk_h.value, N, H, dev,
                                                                          o.get("gamma_h"), o.get("beta_h"))
        _call("gnm_node_bwd_apply", N, H, _ptr(s.z), _ptr(s.stat_h), _ptr(bstat_h), _ptr(prm.gamma_h),
              _ptr(gh_out), _ptr(s.inv_f), _ptr(s.inv_b), _ptr(gP), _ptr(Q), st)
        return gP, Q

    need_f = lib.gnm_edge_bwd_fused_workspace_bytes()
    need_p = lib.gnm_node_proj_bwd_workspace_bytes(5 * H)
    need_t = lib.gnm_tn128_workspace_bytes()
    i = L - 1
    prm, s = ensure(i)
    gP, Q = node(i, gh)
    nblk = C.c_int(0)
    UT = None                   # two-sided sweep: the raw by-source sums [Us | Ts] of the current layer (None: src(i) forms them)
    if plan is None:
        Ud, Td = torch.empty(N, H, **f32), torch.empty(N, H, **f32)
        _call("gnm_edge_bwd_dst", N, E, H, _ptr(s.e_out), _ptr(s.t), _ptr(s.stat_e), _ptr(ge), _ptr(s.P),
              _ptr(Q), _ptr(s.hf), _ptr(s.hb), _ptr(idx["isrc"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(Ud), _ptr(Td),
              _ptr(sc.partials), C.byref(nblk), st)
    else:                       # the top layer on the same two-sided sweep as the chained kernel (no layer above)
        UT = torch.empty(N, 2 * H, **f32)
        DT = torch.empty(N, 2 * H, **f32)
        Ud, Td = DT[:, :H], DT[:, H:]
        ws = sc.ws(max(need_p, need_f, need_t))
        _call("gnm_edge_bwd_top", N, E, H, _ptr(ge), _ptr(s.e_out), _ptr(s.t), _ptr(s.stat_e), _ptr(s.P), _ptr(Q), _ptr(s.hf),
              _ptr(s.hb), _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(Ud), _ptr(Td),
              _ptr(sc.partials), _ptr(plan["sinfo"]), plan["nodes_per_block"], _ptr(UT), C.byref(nblk), _ptr(ws), need_f, st)
        _call("gnm_edge_bwd_src_fix", plan["nfix"], _ptr(plan["fix_nodes"]), N, E, H, _ptr(s.e_out), _ptr(s.t),
              _ptr(s.stat_e), _ptr(ge), _ptr(Q), _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]),
              _ptr(gP), _ptr(UT), st)
    if current().ACTIVATIONS == "lean":
        s.P = None              # as below: only the by-destination pass reads the rebuilt P
    # not in the lean mode: the deferred kernel keeps its gP (one [E,H]-sized tensor) alive one layer longer
    side = _side_stream(dev) if (current().TN_SIDE and _prof is None and current().ACTIVATIONS != "lean") else None
    main = torch.cuda.current_stream()
    fusedn = current().NODE_FUSED and plan is not None      # the node side without node_bgrad / node_bwd_stats launches (see NODE_FUSED)
    at_now = tn_at(N) == "now"          # when the deferred weight-gradient kernel is launched (TN_AT; "auto": by graph size)
    pending = None              # (gP, h_in, gW5, gb5) of the layer above: its weight-gradient kernel, not yet launched
    pending2 = None             # the same, when only its first launch (TN_SPLIT) has been issued
    held: List[torch.Tensor] = []   # what the side stream is reading; dropped only after the main stream has waited for it

    def side_begin():
        main.wait_stream(side)      # the previous deferred kernel ended a layer ago: free, and makes `held` safe to drop
        held.clear()                # (no record_stream: blocks parked on a side-stream event made the allocator
        side.wait_stream(main)      #  fall back to hipMalloc / hipFree in the lean mode: 3x the step time)

    while True:
        prm, s = prms[i], saved[i]
        o = outs[i] or {}
        g = grads[i]
        bstat_e, g["gamma_e"], g["beta_e"] = bn_bwd_finalize(sc.partials, nblk.value, E, H, dev, o.get("gamma_e"), o.get("beta_e"))
        g["W5"], g["b5"] = tgt(i, "W5", 5 * H, H), tgt(i, "b5", 5 * H)
        gh_in = torch.empty(N, H, **f32)
        ws = sc.ws(max(need_p, need_f, need_t))
        nblk_h = None
        if fusedn:
            # ---- round 5: tn34(i) [forms gB1h | gB2h from the raw sums, writes them to gP, their weight gradient] ->
            #      nn(i) [+ the BatchNorm_h backward sums of layer i-1 in its epilogue];  the weight gradient of the other three
            #      column groups on the side stream beside the HBM-bound BatchNorm_h backward of layer i-1
            sc3 = scratch(dev, "tn")
            if side is not None and pending is not None:        # tn012 of the layer above, deferred (TN_AT = "next")
                pgP, ph, pW, pb = pending
                side_begin()
                tn128(N, pgP, 5 * H, 3, ph, pW, pb, sc3.partials, sc3.ws(need_t), need_t, stream=side)
                held.extend((pgP, ph))
                pending = None
            _call("gnm_tn128_bgrad", N, H, _ptr(UT), _ptr(Ud), _ptr(Td), Ud.stride(0), _ptr(s.stat_e), _ptr(bstat_e),
                  _ptr(prm.gamma_e), _ptr(idx["in_ptr"]), _ptr(idx["out_ptr"]), _ptr(gP),
                  _ptr(s.h_in), _ptr(g["W5"][3 * H:]), _ptr(g["b5"][3 * H:]),
                  _ptr(sc.partials), _ptr(ws), need_t, st)
            del Ud, Td, Q
            UT = None
            if i > 0:
                s_j = ensure(i - 1)[1]
                nblk_h = C.c_int(0)
                _call("gnm_node_proj_bwd_nn_stats", N, H, 5 * H, _ptr(gP), _ptr(prm.W5), _ptr(gh), _ptr(gh_in), _ptr(s_j.z),
                      _ptr(s_j.stat_h), _ptr(sc.partials), C.byref(nblk_h), _ptr(ws), need_p, st)
            else:
                _call("gnm_node_proj_bwd_nn", N, H, 5 * H, _ptr(gP), _ptr(prm.W5), _ptr(gh), _ptr(gh_in), _ptr(ws), need_p, st)
            if side is not None and at_now:             # tn012(i) right away, beside node(i-1)'s [N,H] passes
                side_begin()
                tn128(N, gP, 5 * H, 3, s.h_in, g["W5"], g["b5"], sc3.partials, sc3.ws(need_t), need_t, stream=side)
                held.extend((gP, s.h_in))
            elif side is not None and i > 0:
                pending = (gP, s.h_in, g["W5"], g["b5"])
            else:       # no side stream (lean activations, per-op timing) or the last iteration: the same launch on this stream
                tn128(N, gP, 5 * H, 3, s.h_in, g["W5"], g["b5"], sc.partials if nblk_h is None else sc3.partials,
                      sc.ws(max(need_p, need_f, need_t)) if nblk_h is None else sc3.ws(need_t), need_t, tag="gnm_tn128[3]")
        else:
            src_cap = 0
            if pending is not None:
                # the matrix-bound weight gradient of the layer above on the side stream, beside this layer's HBM-bound
                # by-source pass (see TN_SIDE)
                pgP, ph, pW, pb = pending
                side_begin()
                sc3 = scratch(dev, "tn")
                if current().TN_SPLIT and UT is not None:
                    tn128(N, pgP[:, 3 * H:], 5 * H, 2, ph, pW[3 * H:], pb[3 * H:], sc3.partials, sc3.ws(need_t), need_t, stream=side)
                    pending2 = pending
                else:
                    ws3 = sc3.ws(need_p)
                    _lib.check(lib.gnm_node_proj_bwd_tn(N, H, 5 * H, _ptr(pgP), _ptr(ph), _ptr(pW), _ptr(pb), _ptr(sc3.partials),
                                                        _ptr(ws3), need_p, current().TN_SIDE_CAP, C.c_void_p(side.cuda_stream)),
                               "gnm_node_proj_bwd_tn")
                held.extend((pgP, ph))
                pending = None
                src_cap = current().SRC_SIDE_CAP
            if UT is None:
                _call("gnm_edge_bwd_src", N, E, H, _ptr(s.e_out), _ptr(s.t), _ptr(s.stat_e), _ptr(bstat_e),
                      _ptr(prm.gamma_e), _ptr(ge), _ptr(Q), _ptr(idx["in_ptr"]), _ptr(idx["out_ptr"]),
                      _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(Ud), _ptr(Td), _ptr(gP), src_cap, st)
            else:       # the sums are there (chain_src + fix): only the conversion through m1, m2 is left
                _call("gnm_node_bgrad", N, H, _ptr(s.stat_e), _ptr(bstat_e), _ptr(prm.gamma_e), _ptr(idx["in_ptr"]),
                      _ptr(idx["out_ptr"]), _ptr(UT), _ptr(Ud), _ptr(Td), Ud.stride(0), _ptr(gP), st)
            split2 = UT is not None
            del Ud, Td, Q
            UT = None
            if not (side is not None and at_now):
                _call("gnm_node_proj_bwd_nn", N, H, 5 * H, _ptr(gP), _ptr(prm.W5), _ptr(gh), _ptr(gh_in), _ptr(ws), need_p, st)
            if pending2 is not None:        # the other three column groups of the layer above, behind this layer's nn
                pgP, ph, pW, pb = pending2
                side.wait_stream(main)
                sc3 = scratch(dev, "tn")
                tn128(N, pgP, 5 * H, 3, ph, pW, pb, sc3.partials, sc3.ws(need_t), need_t, stream=side)
                pending2 = None
            if side is not None and at_now:
                side_begin()
                sc3 = scratch(dev, "tn")
                ws3 = sc3.ws(need_p)
                _lib.check(lib.gnm_node_proj_bwd_tn(N, H, 5 * H, _ptr(gP), _ptr(s.h_in), _ptr(g["W5"]), _ptr(g["b5"]), _ptr(sc3.partials),
                                                    _ptr(ws3), need_p, current().TN_SIDE_CAP, C.c_void_p(side.cuda_stream)),
                           "gnm_node_proj_bwd_tn")
                held.extend((gP, s.h_in))
                _call("gnm_node_proj_bwd_nn", N, H, 5 * H, _ptr(gP), _ptr(prm.W5), _ptr(gh), _ptr(gh_in), _ptr(ws), need_p, st)
            elif side is not None and i > 0:
                pending = (gP, s.h_in, g["W5"], g["b5"])
            elif side is None and i > 0 and current().TN_SPLIT and split2:
                # no side stream (lean activations, per-op timing): the same two launches the deferred path issues, back to back --
                # the row partition of a launch depends on its column-group count, and the two modes must stay bit-identical
                ws_t = sc.ws(max(need_p, need_f, need_t))
                tn128(N, gP[:, 3 * H:], 5 * H, 2, s.h_in, g["W5"][3 * H:], g["b5"][3 * H:], sc.partials, ws_t, need_t,
                      tag="gnm_node_proj_bwd_tn")
                tn128(N, gP, 5 * H, 3, s.h_in, g["W5"], g["b5"], sc.partials, ws_t, need_t, tag="gnm_node_proj_bwd_tn")
            else:
                _call("gnm_node_proj_bwd_tn", N, H, 5 * H, _ptr(gP), _ptr(s.h_in), _ptr(g["W5"]), _ptr(g["b5"]),
                      _ptr(sc.partials), _ptr(ws), need_p, 0, st)
        del gP
        gh = gh_in
        g["W3"], g["b3"] = tgt(i, "W3", H, H), tgt(i, "b3", H)
        if i == 0:
            _call("gnm_edge_bwd_fused", E, H, _ptr(ge), _ptr(ge), _ptr(s.t), _ptr(s.e_in), _ptr(s.stat_e), _ptr(bstat_e),
                  _ptr(prm.gamma_e), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]), _ptr(sc.partials), _ptr(ws), need_f, st)
            saved[0] = None
            if side is not None:
                main.wait_stream(side)
                held.clear()
            break
        j = i - 1
        prm_j, s_j = ensure(j)
        gP, Q = node(j, gh, nblk_h)
        if plan is None:
            Ud, Td = torch.empty(N, H, **f32), torch.empty(N, H, **f32)
            _call("gnm_edge_bwd_chain", N, E, H, _ptr(ge), _ptr(ge), _ptr(s.t), _ptr(s.e_in), _ptr(s.stat_e), _ptr(bstat_e),
                  _ptr(prm.gamma_e), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]), _ptr(sc2.partials),
                  _ptr(s_j.t), _ptr(s_j.stat_e), _ptr(s_j.P), _ptr(Q), _ptr(s_j.hf), _ptr(s_j.hb),
                  _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(Ud), _ptr(Td), _ptr(sc.partials),
                  C.byref(nblk), _ptr(ws), need_f, st)
        else:
            UT = torch.empty(N, 2 * H, **f32)
            DT = torch.empty(N, 2 * H, **f32)       # [Ud | Td] in one array (the run-sum variant stores both through one buffer)
            Ud, Td = DT[:, :H], DT[:, H:]
            _call("gnm_edge_bwd_chain_src", N, E, H, _ptr(ge), _ptr(ge), _ptr(s.t), _ptr(s.e_in), _ptr(s.stat_e), _ptr(bstat_e),
                  _ptr(prm.gamma_e), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]), _ptr(sc2.partials),
                  _ptr(s_j.t), _ptr(s_j.stat_e), _ptr(s_j.P), _ptr(Q), _ptr(s_j.hf), _ptr(s_j.hb),
                  _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(Ud), _ptr(Td), _ptr(sc.partials),
                  _ptr(plan["sinfo"]), plan["nodes_per_block"], _ptr(UT), C.byref(nblk), _ptr(ws), need_f, st)
            # the sources the plan does not serve (chunk boundaries, repeat edges, no out-edges): ge holds ge_tot(j) now
            _call("gnm_edge_bwd_src_fix", plan["nfix"], _ptr(plan["fix_nodes"]), N, E, H, _ptr(s_j.e_out), _ptr(s_j.t),
                  _ptr(s_j.stat_e), _ptr(ge), _ptr(Q), _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]),
                  _ptr(gP), _ptr(UT), st)
        saved[i] = None         # release layer i's activations
        if current().ACTIVATIONS == "lean":
            s_j.P = None        # rebuilt for the by-destination pass only; nothing after it reads P
        i = j
    return gh, ge, grads


def ln_chain_eligible(H: int, batch_norm: bool) -> bool:
    """The chained LayerNorm backward (round 6): H = 128, split matmul modes, the two-sided LayerNorm sweep switched on."""
    o = current()
    return (not batch_norm) and o.CHAIN and o.FUSED and o.TWO_SIDED and o.LN_SWEEP and H == 128 and _lib.split_mode()


def layers_backward_chained_ln(idx, N: int, E: int, H: int, P: Dict[str, torch.Tensor], L: int, saved: List[LayerSaved], gh, ge,
                               outs: List[Optional[Dict[str, torch.Tensor]]], plan: dict, lnw: int):
    """Backward of an L-layer LayerNorm stack (batch_norm=False), chained like layers_backward_chained.  LayerNorm has no global
    statistics, so the schedule is shorter than BatchNorm's -- no finalisation between a layer's passes, no conversion of raw sums:
        node(L-1), sweep(L-1) [+ fix];   then for i = L-1 .. 0:   nn(i), tn(i) [side stream],
                                              i > 0:  node(i-1), CHAIN[fused(i) with gt(i) given + sweep(i-1)] [+ fix(i-1)]
                                              i = 0:  fused(0) with gt(0) given
    The sweep of layer i writes gt(i) once; the chained kernel of the next iteration reads it back as layer i's given gt (6 [E,H]
    streams per layer where the layer-by-layer schedule moves 9).  Returns (gh_in of layer 0, ge_in of layer 0, [grads dict per layer])."""
    lib = _lib.load()
    dev = gh.device
    sc, sc2 = scratch(dev), scratch(dev, "side")
    st = _stream()
    f32 = dict(dtype=torch.float32, device=dev)
    grads: List[Dict[str, torch.Tensor]] = [dict() for _ in range(L)]
    prms = [None] * L
    need_f = lib.gnm_edge_bwd_fused_workspace_bytes()
    need_p = lib.gnm_node_proj_bwd_workspace_bytes(5 * H)

    def tgt(i, key, *shape):
        o = outs[i] or {}
        return o[key] if key in o else torch.empty(*shape, **f32)

    def ensure(i):
        if prms[i] is None:
            prms[i] = layer_params(P, i)
        s = saved[i]
        _same_matmul_mode(s)
        if s.P is None or s.t is None:      # "lean" activations
            s.P, s.t = _proj_and_t(idx, N, E, H, prms[i], s.h_in, s.e_in, C.c_int(0))
        return prms[i], s

    def node(i, gh_out):
        prm, s = ensure(i)
        o = outs[i] or {}
        gP = torch.empty(N, 5 * H, **f32)
        Q = torch.empty(N, 4 * H, **f32)
        nb = C.c_int(0)
        _call("gnm_ln_node_bwd", N, H, _ptr(s.z), _ptr(prm.gamma_h), _ptr(prm.beta_h), _ptr(gh_out), _ptr(s.hf),
              _ptr(s.inv_f), _ptr(s.hb), _ptr(s.inv_b), _ptr(gP), _ptr(Q), _ptr(sc.partials), C.byref(nb), lnw, st)
        _, grads[i]["gamma_h"], grads[i]["beta_h"] = bn_bwd_finalize(sc.partials, nb.value, N, H, dev, o.get("gamma_h"), o.get("beta_h"))
        return gP, Q

    def fix_and_finalize(j, s_j, gt_j, Q_j, gP_j, nblk):
        o = outs[j] or {}
        _call("gnm_ln_edge_bwd_src_fix", plan["nfix"], _ptr(plan["fix_nodes"]), N, E, H, _ptr(s_j.e_out), _ptr(gt_j), _ptr(Q_j),
              _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]), _ptr(idx["out_dst"]), _ptr(gP_j), st)
        _, grads[j]["gamma_e"], grads[j]["beta_e"] = bn_bwd_finalize(sc.partials, nblk.value, E, H, dev, o.get("gamma_e"), o.get("beta_e"))

    i = L - 1
    prm, s = ensure(i)
    gP, Q = node(i, gh)
    gt = torch.empty(E, H, **f32)
    nblk = C.c_int(0)
    ws = sc.ws(max(need_f, need_p))
    _call("gnm_ln_edge_bwd_top", N, E, H, _ptr(ge), _ptr(s.e_out), _ptr(s.t), _ptr(prm.gamma_e), _ptr(prm.beta_e), lnw,
          _ptr(s.P), _ptr(Q), _ptr(s.hf), _ptr(s.hb), _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(gt),
          _ptr(sc.partials), _ptr(plan["sinfo"]), plan["nodes_per_block"], C.byref(nblk), _ptr(ws), need_f, st)
    fix_and_finalize(i, s, gt, Q, gP, nblk)
    del Q
    use_side = current().TN_SIDE and _prof is None and current().ACTIVATIONS != "lean"
    while True:
        prm, s = prms[i], saved[i]
        g = grads[i]
        g["W5"], g["b5"] = tgt(i, "W5", 5 * H, H), tgt(i, "b5", 5 * H)
        gh_in = torch.empty(N, H, **f32)
        ws = sc.ws(max(need_f, need_p))
        _call("gnm_node_proj_bwd_nn", N, H, 5 * H, _ptr(gP), _ptr(prm.W5), _ptr(gh), _ptr(gh_in), _ptr(ws), need_p, st)
        if use_side:
            side = side_begin(dev)
            sc3 = scratch(dev, "tn")
            ws3 = sc3.ws(need_p)
            _lib.check(lib.gnm_node_proj_bwd_tn(N, H, 5 * H, _ptr(gP), _ptr(s.h_in), _ptr(g["W5"]), _ptr(g["b5"]), _ptr(sc3.partials),
                                                _ptr(ws3), need_p, current().TN_SIDE_CAP, C.c_void_p(side.cuda_stream)), "gnm_node_proj_bwd_tn")
            _side_held[torch.device(dev)].extend((gP, s.h_in, g["W5"], g["b5"]))
        else:
            _call("gnm_node_proj_bwd_tn", N, H, 5 * H, _ptr(gP), _ptr(s.h_in), _ptr(g["W5"]), _ptr(g["b5"]), _ptr(sc.partials), _ptr(ws),
                  need_p, 0, st)
        del gP
        gh = gh_in
        g["W3"], g["b3"] = tgt(i, "W3", H, H), tgt(i, "b3", H)
        if i == 0:
            _call("gnm_edge_bwd_fused_gt", E, H, _ptr(ge), _ptr(ge), _ptr(gt), _ptr(s.e_in), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]),
                  _ptr(sc.partials), _ptr(ws), need_f, st)
            saved[0] = None
            side_drain(dev)
            break
        j = i - 1
        prm_j, s_j = ensure(j)
        gP, Q = node(j, gh)
        gt_j = torch.empty(E, H, **f32)
        ws = sc.ws(max(need_f, need_p))
        _call("gnm_ln_edge_bwd_chain", N, E, H, _ptr(ge), _ptr(gt), _ptr(s.e_in), _ptr(prm.W3), _ptr(g["W3"]), _ptr(g["b3"]),
              _ptr(sc2.partials), _ptr(s_j.t), _ptr(prm_j.gamma_e), _ptr(prm_j.beta_e), lnw, _ptr(s_j.P), _ptr(Q), _ptr(s_j.hf),
              _ptr(s_j.hb), _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["in_ptr"]), _ptr(gP), _ptr(gt_j), _ptr(sc.partials),
              _ptr(plan["sinfo"]), plan["nodes_per_block"], C.byref(nblk), _ptr(ws), need_f, st)
        fix_and_finalize(j, s_j, gt_j, Q, gP, nblk)
        del Q
        saved[i] = None         # release layer i's activations
        if current().ACTIVATIONS == "lean":
            s_j.P = None
        gt = gt_j
        i = j
    return gh, ge, grads


# ---------------------------------------------------------------------------------------
# predictor (score_predictor.py:12-25), split-W1 form
# ---------------------------------------------------------------------------------------

@dataclass
class PredSaved:
    x: torch.Tensor = None
    e: torch.Tensor = None
    hid: torch.Tensor = None     # pre-activation [E,HS]; overwritten by its gradient in backward
    W1sd: torch.Tensor = None
    opts: "Options" = None


@on_device_of(lambda idx, N, E, H, W1, b1, W2, b2, x, *a, **k: x)
@_scoped()
def predictor_forward(idx, N, E, H, W1, b1, W2, b2, x, e, save: bool):
    """scores (caller edge-id order, [E,1]) from internal-order x [N,H], e [E,H]."""
    lib = _lib.load()
    dev = x.device
    HS = W1.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    W1sd = torch.cat((W1[:, :H], W1[:, H:2 * H]), 0).contiguous()     # [2HS, H]
    Pn = torch.empty(N, 2 * HS, **f32)
    gemm(NT, x, W1sd, Pn)
    scores = torch.empty(E, 1, **f32)
    if (H == 128 or (H == 256 and current().WIDE_FUSED)) and HS == 64 and current().FUSED:
        # one pass over e: hid GEMM + gathers + relu + W2 dot; hid is only written when backward needs it
        hid = torch.empty(E, HS, **f32) if save else None
        need = lib.gnm_predictor_fused_workspace_bytes()
        ws = scratch(dev).ws(need)
        W1e = W1[:, 2 * H:]
        _call("gnm_predictor_fused_fwd", E, H, HS, _ptr(e), _ptr(W1e), W1.stride(0), _ptr(b1), _ptr(Pn),
              _ptr(idx["isrc"]), _ptr(idx["idst"]), _ptr(idx["perm"]), _ptr(W2), _ptr(b2), _ptr(hid), _ptr(scores),
              _ptr(ws), need, _stream())
    else:
        hid = torch.empty(E, HS, **f32)
        gemm(NT, e, W1[:, 2 * H:], hid, bias=b1)
        _call("gnm_predictor_score_fwd", E, HS, _ptr(hid), _ptr(Pn), _ptr(idx["isrc"]), _ptr(idx["idst"]),
                                               _ptr(W2), _ptr(b2), _ptr(idx["perm"]), _ptr(scores), _stream())
    saved = PredSaved(x=x, e=e, hid=hid, W1sd=W1sd, opts=current()) if save else None
    return scores, saved


@on_device_of(lambda idx, N, E, H, W1, W2, s, gscores, *a, **k: gscores)
@_scoped(6)
def predictor_backward(idx, N, E, H, W1, W2, s: PredSaved, gscores, out: Optional[Dict[str, torch.Tensor]] = None):
    """Returns (gx [N,H], ge [E,H] fresh buffer, grads dict W1,b1,W2,b2); `out` as in layer_backward."""
    out = out or {}
    lib = _lib.load()
    dev = s.x.device
    sc = scratch(dev)
    HS = W1.shape[0]
    st = _stream()
    nblk = C.c_int(0)
    f32 = dict(dtype=torch.float32, device=dev)
    g = {}
    gscores = _f32c(gscores.reshape(-1))
    ghid = s.hid   # in place
    fused = (H == 128 or (H == 256 and current().WIDE_FUSED)) and HS == 64 and current().FUSED
    gW1 = out["W1"] if "W1" in out else torch.empty(HS, 3 * H, **f32)
    if fused:
        # one pass: ghid (in place), ge = ghid W1e, gW1e, and the gW2 / gb1 / gb2 column sums
        ge = torch.empty(E, H, **f32)
        gW1e = torch.empty(HS, H, **f32)
        gsums = torch.empty(3 * HS, **f32)
        need = lib.gnm_predictor_fused_workspace_bytes()
        ws = sc.ws(need)
        _call("gnm_predictor_fused_bwd", E, H, HS, _ptr(ghid), _ptr(gscores), _ptr(idx["perm"]), _ptr(W2), _ptr(s.e),
              _ptr(W1[:, 2 * H:]), W1.stride(0), _ptr(ge), _ptr(gW1e), _ptr(gsums), _ptr(sc.partials), _ptr(ws), need, st)
        def keep(key, src):
            if key in out:
                out[key].copy_(src.reshape(out[key].shape))
                return out[key]
            return src.clone()
        g["W2"] = keep("W2", gsums[0:HS].reshape(1, HS))
        g["b1"] = keep("b1", gsums[HS:2 * HS])
        g["b2"] = keep("b2", gsums[2 * HS:2 * HS + 1])
        gW1[:, 2 * H:] = gW1e
    else:
        _call("gnm_predictor_score_bwd", E, HS, _ptr(ghid), _ptr(gscores), _ptr(W2), _ptr(idx["perm"]),
                                               _ptr(sc.partials), C.byref(nblk), st)
        red = torch.empty(2, HS, **f32)
        _call("gnm_reduce_partials", _ptr(sc.partials), nblk.value, 2, HS, _ptr(red), st)
        g["W2"] = out["W2"].copy_(red[0:1]) if "W2" in out else red[0:1].clone()
        g["b2"] = out["b2"].copy_(red[1, 0:1]) if "b2" in out else red[1, 0:1].clone()
        g["b1"] = colsum(ghid, out.get("b1"))
    gPn = torch.empty(N, 2 * HS, **f32)
    _call("gnm_seg_sum_rows", N, HS, _ptr(ghid), _ptr(idx["out_ptr"]), _ptr(idx["out_pos"]),
                                    _ptr(gPn), 2 * HS, st)
    _call("gnm_seg_sum_rows", N, HS, _ptr(ghid), _ptr(idx["in_ptr"]), C.c_void_p(0),
                                    _ptr(gPn[:, HS:]), 2 * HS, st)
    if fused:
        # [x^T gPs | x^T gPd] in one W-free TN pass (x's 128-column groups against the 128-wide gPn), then transpose the H x 128 result
        xt = torch.empty(H, 2 * HS, **f32)
        junk = torch.empty(H, **f32)
        need = lib.gnm_tn128_workspace_bytes()
        ws = sc.ws(need)
        _call("gnm_tn128", N, _ptr(s.x), H, H // 128, _ptr(gPn), _ptr(xt), _ptr(junk), _ptr(sc.partials), _ptr(ws), need, st)
        gW1[:, :H] = xt[:, :HS].t()
        gW1[:, H:2 * H] = xt[:, HS:].t()
    else:
        gemm(TN, gPn[:, :HS], s.x, gW1[:, :H])
        gemm(TN, gPn[:, HS:], s.x, gW1[:, H:2 * H])
        gemm(TN, ghid, s.e, gW1[:, 2 * H:])
    g["W1"] = gW1
    gx = torch.empty(N, H, **f32)
    gemm(NN, gPn, s.W1sd, gx)
    if not fused:
        ge = torch.empty(E, H, **f32)
        gemm(NN, ghid, W1[:, 2 * H:], ge)
    return gx, ge, g


# ---------------------------------------------------------------------------------------
# whole model (full_graph.py:22-29)
# ---------------------------------------------------------------------------------------

@dataclass
class ModelSaved:
    pe: torch.Tensor = None
    e_int: torch.Tensor = None
    a1: torch.Tensor = None
    e_raw: torch.Tensor = None
    layers: List[LayerSaved] = field(default_factory=list)
    pred: PredSaved = None
    opts: "Options" = None
    matmul: str = None


def stacked(ts):
    """cat(ts, 0) -- as a VIEW when the tensors already sit back to back in one storage (parameters and gradients
    flattened by models.flatten_parameters / dp.FlatGradients), which saves the copy kernel per layer and pass."""
    t0 = ts[0]
    adjacent = all(t.is_contiguous() and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] for t in ts) and all(
        a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
        and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr() for a, b in zip(ts[:-1], ts[1:]))
    if not adjacent:
        return torch.cat(ts, 0)
    rows = sum(t.shape[0] for t in ts)
    size = (rows,) + tuple(t0.shape[1:])
    stride = t0.stride()
    return torch.as_strided(t0, size, stride)


def layer_params(P: Dict[str, torch.Tensor], i: int) -> LayerParams:
    p = f"gnn.convs.{i}."
    return LayerParams(
        W5=stacked([P[p + k + ".weight"] for k in LIN5]),
        b5=stacked([P[p + k + ".bias"] for k in LIN5]),
        W3=P[p + "B_3.weight"], b3=P[p + "B_3.bias"],
        gamma_e=P[p + "bn_e.weight"], beta_e=P[p + "bn_e.bias"],
        gamma_h=P[p + "bn_h.weight"], beta_h=P[p + "bn_h.bias"])


@on_device_of(lambda graph, e_raw, pe, *a, **k: pe)
@_scoped()
def model_forward(graph, e_raw, pe, P: Dict[str, torch.Tensor], num_layers: int, save: bool, batch_norm: bool = True,
                  ln_width: Optional[int] = None):
    """GraphGatedGCNModel.forward.  e_raw [E,edge_features] in edge-id order, pe [N,nb_pos_enc+2].
    Returns (scores [E,1] in edge-id order, ModelSaved or None).  ln_width: see layer_forward."""
    lib = _lib.load()
    dev = pe.device
    _chk_dev(e_raw, pe)
    idx = graph.index(dev)
    N, E = graph.num_nodes(), graph.num_edges()
    H = P["linear_pe.weight"].shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    pe = node_rows_in(idx, _f32c(pe))         # caller's node numbering -> internal (graph.py "Node numbering")
    e_raw = _f32c(e_raw)
    # encoders                                                            (full_graph.py:23-26)
    h = torch.empty(N, H, **f32)
    gemm(NT, pe, P["linear_pe.weight"], h, bias=P["linear_pe.bias"])
    Fe, Q = e_raw.shape[1], P["linear1_edge.weight"].shape[0]
    fused_enc = current().FUSED and (H == 128 or (H == 256 and current().WIDE_FUSED)) and Fe == 2 and Q == 16
    e = torch.empty(E, H, **f32)
    e_int = a1 = None
    if fused_enc:
        _call("gnm_edge_encoder_fwd", E, H, Fe, Q, _ptr(e_raw), _ptr(idx["perm"]), _ptr(P["linear1_edge.weight"]),
              _ptr(P["linear1_edge.bias"]), _ptr(P["linear2_edge.weight"]), _ptr(P["linear2_edge.bias"]),
              _ptr(e), _stream())
    else:
        e_int = torch.empty(E, Fe, **f32)
        _call("gnm_gather_rows_f32", E, Fe, _ptr(e_raw), _ptr(idx["perm"]), _ptr(e_int), _stream())
        a1 = torch.empty(E, Q, **f32)
        gemm(NT, e_int, P["linear1_edge.weight"], a1, bias=P["linear1_edge.bias"], relu=True)
        gemm(NT, a1, P["linear2_edge.weight"], e, bias=P["linear2_edge.bias"])
    ms = ModelSaved(pe=pe, e_int=e_int, a1=a1, e_raw=e_raw, opts=current(), matmul=_lib.get_matmul_mode()) if save else None
    plan2 = graph.sweep_plan(dev, GATE2_WG) if (current().TWO_SIDED_FWD and sweep_width(H, batch_norm) and hasattr(graph, "sweep_plan")) else None
    for i in range(num_layers):
        h, e, ls = layer_forward(idx, N, E, H, layer_params(P, i), h, e, save, batch_norm, plan=plan2, ln_width=ln_width)
        if save:
            ms.layers.append(ls)
    scores, ps = predictor_forward(idx, N, E, H, P["predictor.W1.weight"], P["predictor.W1.bias"],
                                   P["predictor.W2.weight"], P["predictor.W2.bias"], h, e, save)
    if save:
        ms.pred = ps
    return scores, ms


def grad_targets(out: Dict[str, torch.Tensor], i: int) -> Optional[Dict[str, torch.Tensor]]:
    """The write-into map of layer_backward for layer i from per-parameter gradient tensors, if the five stacked
    weights / biases are back to back (they are in dp.FlatGradients over a flattened model); else None."""
    p = f"gnn.convs.{i}."
    w = [out[p + k + ".weight"] for k in LIN5]
    b = [out[p + k + ".bias"] for k in LIN5]
    W5, b5 = stacked(w), stacked(b)
    if W5.data_ptr() != w[0].data_ptr() or b5.data_ptr() != b[0].data_ptr():
        return None
    return {"W5": W5, "b5": b5, "W3": out[p + "B_3.weight"], "b3": out[p + "B_3.bias"],
            "gamma_e": out[p + "bn_e.weight"], "beta_e": out[p + "bn_e.bias"],
            "gamma_h": out[p + "bn_h.weight"], "beta_h": out[p + "bn_h.bias"]}


@on_device_of(lambda graph, P, num_layers, ms, gscores, *a, **k: gscores)
@_scoped(3)
def model_backward(graph, P: Dict[str, torch.Tensor], num_layers: int, ms: ModelSaved, gscores, batch_norm: bool = True,
                   out: Optional[Dict[str, torch.Tensor]] = None, ln_width: Optional[int] = None):
    """Gradients of every parameter (keys = state_dict keys) from d loss / d scores.  With `out` (state_dict key ->
    contiguous tensor of the parameter's shape, e.g. the .grad views of dp.FlatGradients) the kernels write the
    gradients straight into those tensors and the same tensors are returned."""
    dev = ms.pe.device
    _same_matmul_mode(ms)
    idx = graph.index(dev)
    N, E = graph.num_nodes(), graph.num_edges()
    H = P["linear_pe.weight"].shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    G: Dict[str, torch.Tensor] = {}
    tgt = (lambda k, like: out[k]) if out else (lambda k, like: torch.empty_like(like))    # noqa: E731
    pout = {"W1": out["predictor.W1.weight"], "b1": out["predictor.W1.bias"], "W2": out["predictor.W2.weight"],
            "b2": out["predictor.W2.bias"]} if out else None
    gh, ge, gp = predictor_backward(idx, N, E, H, P["predictor.W1.weight"], P["predictor.W2.weight"],
                                    ms.pred, gscores, pout)
    G["predictor.W1.weight"], G["predictor.W1.bias"] = gp["W1"], gp["b1"]
    G["predictor.W2.weight"], G["predictor.W2.bias"] = gp["W2"], gp["b2"]
    ms.pred = None
    louts = [grad_targets(out, i) if out else None for i in range(num_layers)]
    chained = None
    # layer-by-layer backward on the two-sided sweep (the chained schedule's top-layer kernel for every layer): H = 256, and H = 128
    # where the chained schedule does not apply (fp32-MFMA matmul mode, GNM_CHAIN=0)
    plan_w = graph.sweep_plan(dev) if (sweep_width(H, batch_norm) and current().TWO_SIDED and not chain_eligible(H, batch_norm)
                                       and hasattr(graph, "sweep_plan")) else None
    if chain_eligible(H, batch_norm):
        plan = graph.sweep_plan(dev) if current().TWO_SIDED and hasattr(graph, "sweep_plan") else None
        gh, ge, chained = layers_backward_chained(idx, N, E, H, P, num_layers, ms.layers, gh, ge, louts, plan)
    elif ln_chain_eligible(H, batch_norm) and plan_w is not None:
        gh, ge, chained = layers_backward_chained_ln(idx, N, E, H, P, num_layers, ms.layers, gh, ge, louts, plan_w,
                                                     H if ln_width is None else int(ln_width))
    for i in reversed(range(num_layers)):
        p = f"gnn.convs.{i}."
        lout = louts[i]
        if chained is not None:
            gl = chained[i]
        else:
            gh, ge, gl = layer_backward(idx, N, E, H, layer_params(P, i), ms.layers[i], gh, ge, batch_norm, lout, plan=plan_w,
                                        ln_width=ln_width, defer_tn=True)
            ms.layers[i] = None     # release this layer's activations
        for j, k in enumerate(LIN5):
            G[p + k + ".weight"] = gl["W5"][j * H:(j + 1) * H]
            G[p + k + ".bias"] = gl["b5"][j * H:(j + 1) * H]
        if out and lout is None:          # the caller's targets are not stacked: copy the two stacked results over
            for j, k in enumerate(LIN5):
                G[p + k + ".weight"] = out[p + k + ".weight"].copy_(G[p + k + ".weight"])
                G[p + k + ".bias"] = out[p + k + ".bias"].copy_(G[p + k + ".bias"])
            for key, name in (("W3", "B_3.weight"), ("b3", "B_3.bias"), ("gamma_e", "bn_e.weight"), ("beta_e", "bn_e.bias"),
                              ("gamma_h", "bn_h.weight"), ("beta_h", "bn_h.bias")):
                gl[key] = out[p + name].copy_(gl[key])
        G[p + "B_3.weight"], G[p + "B_3.bias"] = gl["W3"], gl["b3"]
        G[p + "bn_e.weight"], G[p + "bn_e.bias"] = gl["gamma_e"], gl["beta_e"]
        G[p + "bn_h.weight"], G[p + "bn_h.bias"] = gl["gamma_h"], gl["beta_h"]
    if chained is None:
        side_drain(dev)             # the deferred weight-gradient launches of layer_backward(defer_tn=True)
    # encoders backward
    lib = _lib.load()
    G["linear_pe.weight"] = tgt("linear_pe.weight", P["linear_pe.weight"])
    G["linear_pe.bias"] = gemm_tn_colsum(gh, ms.pe, G["linear_pe.weight"], out["linear_pe.bias"] if out else None)
    G["linear2_edge.weight"] = tgt("linear2_edge.weight", P["linear2_edge.weight"])
    G["linear1_edge.weight"] = tgt("linear1_edge.weight", P["linear1_edge.weight"])
    if ms.a1 is None:      # fused encoder: every encoder gradient from one pass over ge
        G["linear2_edge.bias"] = tgt("linear2_edge.bias", P["linear2_edge.bias"])
        G["linear1_edge.bias"] = tgt("linear1_edge.bias", P["linear1_edge.bias"])
        need = lib.gnm_edge_encoder_bwd_workspace_bytes()
        ws = scratch(dev).ws(need)
        _call("gnm_edge_encoder_bwd", E, H, ms.e_raw.shape[1], P["linear1_edge.weight"].shape[0], _ptr(ge),
              _ptr(ms.e_raw), _ptr(idx["perm"]), _ptr(P["linear1_edge.weight"]), _ptr(P["linear1_edge.bias"]),
              _ptr(P["linear2_edge.weight"]), _ptr(G["linear1_edge.weight"]), _ptr(G["linear1_edge.bias"]),
              _ptr(G["linear2_edge.weight"]), _ptr(G["linear2_edge.bias"]), _ptr(ws), need, _stream())
    else:
        gemm(TN, ge, ms.a1, G["linear2_edge.weight"])
        G["linear2_edge.bias"] = colsum(ge, out["linear2_edge.bias"] if out else None)
        ga1 = torch.empty_like(ms.a1)
        gemm(NN, ge, P["linear2_edge.weight"], ga1)
        _call("gnm_relu_mask_f32", ga1.numel(), _ptr(ga1), _ptr(ms.a1), _stream())
        gemm(TN, ga1, ms.e_int, G["linear1_edge.weight"])
        G["linear1_edge.bias"] = colsum(ga1, out["linear1_edge.bias"] if out else None)
    return G


@on_device_of(lambda scores, *a, **k: scores)
def bce_with_logits(scores, y, pos_weight: float):
    """(loss [1], dloss/dscores [E,1]) -- train.py:210-211,253-255, fused in one pass."""
    lib = _lib.load()
    _chk_dev(scores, y)
    x = _f32c(scores.reshape(-1))
    y = _f32c(y.reshape(-1))
    E = x.numel()
    sc = scratch(x.device)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    gs = torch.empty(E, 1, dtype=torch.float32, device=x.device)
    _call("gnm_bce_fwd_bwd", E, _ptr(x), _ptr(y), float(pos_weight), _ptr(loss), _ptr(gs),
                                   _ptr(sc.partials), sc.partials.numel() * 8, _stream())
    return loss, gs


_default = Options(**{k: (globals()["_D_" + k] if k not in ("ACTIVATIONS", "TN_AT") else {"ACTIVATIONS": "saved", "TN_AT": "auto"}[k])
                      for k in _OPTION_NAMES}).replace(ACTIVATIONS=_D_ACTIVATIONS, TN_AT=_D_TN_AT)     # environment values are validated
