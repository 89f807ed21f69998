#!/usr/bin/env python3
"""Distance to fp64 of the three matmul modes (include/gnm.h: f32 / bf16x3 / f16x2), kernel by kernel, on operands chosen to
stress the f16x2 scaling: rows and elements spread over many binades, tiny gradients, column groups of very different
magnitude, zero rows / groups, weights with zero / tiny / huge columns.  gpurun_out/f16x2_accuracy.txt
(rel_l2 = |out - ref|_2 / |ref|_2 over the tensor; cw = max |out - ref| / (|x| |W| + |other terms|), the componentwise bound a
fp32 dot product of length 128 meets at ~ 1e-6)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gnnome_assembly_amd import _lib, engine  # noqa: E402

H = 128
MODES = ("f32", "bf16x3", "f16x2")


def weights(rng, rows, cols, axis):
    W = (rng.standard_normal((rows, cols)) / 11).astype(np.float32)
    sl = [slice(None)] * 2
    for i, f in ((7, 0.0), (9, 1e-20), (11, 1e5)):          # a zero, a tiny and a huge output column
        sl[axis] = i
        W[tuple(sl)] *= f
        sl[axis] = slice(None)
    return W


def report(name, outs, ref, scale):
    cells = []
    for mode, o in outs:
        o = o.astype(np.float64)
        cells.append(f"{mode}: rel_l2 {np.linalg.norm(o - ref) / np.linalg.norm(ref):.2e} cw {np.max(np.abs(o - ref) / np.maximum(scale, 1e-300)):.2e}"
                     + ("" if np.isfinite(o).all() else " NONFINITE"))
    return f"  {name:22s} " + " | ".join(cells)


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    rng = np.random.default_rng(0)
    P_ = engine._ptr
    lines = []

    def ws_for(ncols):
        need = lib.gnm_rowtile_workspace_bytes(ncols)
        return engine.scratch(dev).ws(need), need

    # ---- node projections  P = h W5^T + b5   (gnm_node_proj_fwd)
    lines.append("gnm_node_proj_fwd   P = h W5^T + b5   [N,128] x [640,128]^T")
    N = 20011
    for name, mk in (("normal", lambda: rng.standard_normal((N, H))),
                     ("rows over 26 decades", lambda: rng.standard_normal((N, H)) * np.exp(rng.uniform(-30, 30, (N, 1)))),
                     ("elements over 10 decades", lambda: rng.standard_normal((N, H)) * np.exp(rng.uniform(-12, 12, (N, H)))),
                     ("half the rows zero", lambda: rng.standard_normal((N, H)) * (rng.random((N, 1)) < 0.5))):
        hn = mk().astype(np.float32)
        Wn = weights(rng, 5 * H, H, 0)
        bn = rng.standard_normal(5 * H).astype(np.float32)
        h, W5, b5 = (torch.from_numpy(a).to(dev) for a in (hn, Wn, bn))
        ref = hn.astype(np.float64) @ Wn.astype(np.float64).T + bn
        scale = np.abs(hn).astype(np.float64) @ np.abs(Wn).astype(np.float64).T + np.abs(bn)
        outs = []
        for mode in MODES:
            _lib.set_matmul_mode(mode)
            ws, need = ws_for(5 * H)
            out = torch.empty(N, 5 * H, device=dev)
            engine._call("gnm_node_proj_fwd", N, H, 5 * H, P_(h), P_(W5), P_(b5), P_(out), P_(ws), need, engine._stream())
            torch.cuda.synchronize()
            outs.append((mode, out.cpu().numpy()))
        lines.append(report(name, outs, ref, scale))

    # ---- edge t  = e W3^T + b3 + B1h[src] + B2h[dst]   (gnm_edge_t_fused_fwd)
    lines.append("gnm_edge_t_fused_fwd   t = e W3^T + b3 + P[src, 3H:4H] + P[dst, 4H:5H]   [E,128] x [128,128]^T")
    E, Nn = 50021, 9973
    for name, mk in (("normal", lambda: rng.standard_normal((E, H))),
                     ("rows over 26 decades", lambda: rng.standard_normal((E, H)) * np.exp(rng.uniform(-30, 30, (E, 1)))),
                     ("elements over 10 decades", lambda: rng.standard_normal((E, H)) * np.exp(rng.uniform(-12, 12, (E, H))))):
        en = mk().astype(np.float32)
        Wn = weights(rng, H, H, 0)
        bn = rng.standard_normal(H).astype(np.float32)
        Pn = rng.standard_normal((Nn, 5 * H)).astype(np.float32)
        src = rng.integers(0, Nn, E).astype(np.int32)
        dst = np.sort(rng.integers(0, Nn, E)).astype(np.int32)
        e, W3, b3, Pt, s_, d_ = (torch.from_numpy(a).to(dev) for a in (en, Wn, bn, Pn, src, dst))
        ref = en.astype(np.float64) @ Wn.astype(np.float64).T + bn + Pn[src, 3 * H:4 * H] + Pn[dst, 4 * H:5 * H]
        scale = np.abs(en).astype(np.float64) @ np.abs(Wn).astype(np.float64).T + np.abs(bn) + np.abs(Pn[src, 3 * H:4 * H]) + np.abs(Pn[dst, 4 * H:5 * H])
        outs = []
        for mode in MODES:
            _lib.set_matmul_mode(mode)
            ws, need = ws_for(H)
            out = torch.empty(E, H, device=dev)
            part = torch.zeros(4096 * 2 * H, dtype=torch.float64, device=dev)
            nb = C.c_int(0)
            engine._call("gnm_edge_t_fused_fwd", E, H, P_(e), P_(W3), P_(b3), P_(Pt), P_(s_), P_(d_), P_(out), P_(part), C.byref(nb), P_(ws), need,
                         engine._stream())
            torch.cuda.synchronize()
            outs.append((mode, out.cpu().numpy()))
        lines.append(report(name, outs, ref, scale))

    # ---- projection backward  gh_in = gh_out + gP W5   (gnm_node_proj_bwd_nn / _nn_stats: the accumulator runs over five groups)
    lines.append("gnm_node_proj_bwd_nn_stats   gh_in = gh_out + gP W5   [N,640] x [640,128]")
    rep = lambda a: np.repeat(a, H, axis=1)
    for name, mk in (("normal", lambda: rng.standard_normal((N, 5 * H))),
                     ("gradients of 1e-7", lambda: rng.standard_normal((N, 5 * H)) * 1e-7),
                     ("rows over 13 decades", lambda: rng.standard_normal((N, 5 * H)) * np.exp(rng.uniform(-25, 5, (N, 1)))),
                     ("groups 1e-6 .. 1 apart", lambda: rng.standard_normal((N, 5 * H)) * rep(10.0 ** rng.integers(-6, 1, (N, 5)))),
                     ("groups rising 1e-12..1e12", lambda: rng.standard_normal((N, 5 * H)) * rep(np.tile(np.array([1e-12, 1e-6, 1.0, 1e6, 1e12]), (N, 1)))),
                     ("half the groups zero", lambda: rng.standard_normal((N, 5 * H)) * rep(rng.random((N, 5)) < 0.5))):
        gPn = mk().astype(np.float32)
        Wn = weights(rng, 5 * H, H, 1)
        ghn = (rng.standard_normal((N, H)) * np.abs(gPn).max(1, keepdims=True) * 0.1).astype(np.float32)
        zn = rng.standard_normal((N, H)).astype(np.float32)
        statn = np.stack([np.zeros(H), np.ones(H), np.ones(H), np.zeros(H)]).astype(np.float32)
        gP, W5, gh, z, stat = (torch.from_numpy(a).to(dev) for a in (gPn, Wn, ghn, zn, statn))
        ref = gPn.astype(np.float64) @ Wn.astype(np.float64) + ghn
        scale = np.abs(gPn).astype(np.float64) @ np.abs(Wn).astype(np.float64) + np.abs(ghn)
        outs = []
        for mode in MODES:
            _lib.set_matmul_mode(mode)
            ws, need = ws_for(5 * H)
            out = torch.empty(N, H, device=dev)
            if mode == "f32":
                engine._call("gnm_node_proj_bwd_nn", N, H, 5 * H, P_(gP), P_(W5), P_(gh), P_(out), P_(ws), need, engine._stream())
            else:
                part = torch.zeros(4096 * 2 * H, dtype=torch.float64, device=dev)
                nb = C.c_int(0)
                engine._call("gnm_node_proj_bwd_nn_stats", N, H, 5 * H, P_(gP), P_(W5), P_(gh), P_(out), P_(z), P_(stat), P_(part), C.byref(nb), P_(ws),
                             need, engine._stream())
            torch.cuda.synchronize()
            outs.append((mode, out.cpu().numpy()))
        lines.append(report(name, outs, ref, scale))
    _lib.set_matmul_mode(_lib.DEFAULT_MATMUL_MODE)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "f16x2_accuracy.txt"), "w").write(__doc__.strip() + "\n\n" + "\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
