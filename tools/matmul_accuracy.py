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
    wide(dev, lib, rng, lines)
    _lib.set_matmul_mode(_lib.DEFAULT_MATMUL_MODE)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "f16x2_accuracy.txt"), "w").write(__doc__.strip() + "\n\n" + "\n".join(lines) + "\n")
    print("\n".join(lines))


def wide(dev, lib, rng, lines):
    """Round 6: the kernels of the reference's default width, H = 256 -- the fused t kernel (edge_t32_h256p_k), the
    fused gt + NN kernel (edge_gt_nn_h256_k) and the general row GEMM / weight-gradient kernels the 256-wide node side runs on
    (gemm_rows_b3_k NT K = 256, NN K = 1280; tn_tr_k classes, 20 k rows), each against fp64; the f32 column is the route the fp32-MFMA
    mode takes for the same product (generic fp32 GEMM + elementwise kernel)."""
    P_ = engine._ptr
    W = 256
    N, E, Nn = 20011, 50021, 9973
    lines.append("")
    lines.append("H = 256 ------------------------------------------------------------------------------------------------")
    # ---- generic row GEMMs
    for title, mode_, shp in (("gemm NT  P = h W5^T   [N,256] x [1280,256]^T", engine.NT, (N, 5 * W, W)),
                              ("gemm NN  gh = gP W5   [N,1280] x [1280,256]", engine.NN, (N, W, 5 * W)),
                              ("gemm TN  gW5 = gP^T h   [N,1280]^T x [N,256]", engine.TN, (5 * W, W, N))):
        lines.append(title)
        M_, N_, K_ = shp
        for name, spread in (("normal", None), ("rows over 13 decades", (-25, 5))):
            if mode_ == engine.TN:
                An = rng.standard_normal((K_, M_))
                Bn = rng.standard_normal((K_, N_))
                if spread:
                    An = An * np.exp(rng.uniform(*spread, (K_, 1)))
                ref = An.astype(np.float32).astype(np.float64).T @ Bn.astype(np.float32).astype(np.float64)
                scale = np.abs(An.astype(np.float32)).astype(np.float64).T @ np.abs(Bn.astype(np.float32)).astype(np.float64)
            else:
                An = rng.standard_normal((M_, K_))
                if spread:
                    An = An * np.exp(rng.uniform(*spread, (M_, 1)))
                Bn = (rng.standard_normal((N_, K_) if mode_ == engine.NT else (K_, N_)) / 16)
                a64, b64 = An.astype(np.float32).astype(np.float64), Bn.astype(np.float32).astype(np.float64)
                ref = a64 @ (b64.T if mode_ == engine.NT else b64)
                scale = np.abs(a64) @ (np.abs(b64).T if mode_ == engine.NT else np.abs(b64))
            A, B = (torch.from_numpy(a.astype(np.float32)).to(dev) for a in (An, Bn))
            outs = []
            for mode in MODES:
                _lib.set_matmul_mode(mode)
                out = engine.gemm(mode_, A, B, torch.empty(M_, N_, device=dev))
                torch.cuda.synchronize()
                outs.append((mode, out.cpu().numpy()))
            lines.append(report(name, outs, ref, scale))
    # ---- fused t at H = 256
    lines.append("gnm_edge_t_fused_fwd[256]   t = e W3^T + b3 + B1h[src] + B2h[dst]   [E,256] x [256,256]^T   (f32: gemm NT + gnm_edge_t_stats_fwd)")
    for name, mk in (("normal", lambda: rng.standard_normal((E, W))),
                     ("rows over 26 decades", lambda: rng.standard_normal((E, W)) * np.exp(rng.uniform(-30, 30, (E, 1)))),
                     ("elements over 10 decades", lambda: rng.standard_normal((E, W)) * np.exp(rng.uniform(-12, 12, (E, W))))):
        en = mk().astype(np.float32)
        Wn = weights(rng, W, W, 0)
        bn = rng.standard_normal(W).astype(np.float32)
        Pn = rng.standard_normal((Nn, 5 * W)).astype(np.float32)
        src = rng.integers(0, Nn, E).astype(np.int32)
        dst = np.sort(rng.integers(0, Nn, E)).astype(np.int32)
        e, W3, b3, Pt, s_, d_ = (torch.from_numpy(a).to(dev) for a in (en, Wn, bn, Pn, src, dst))
        ref = en.astype(np.float64) @ Wn.astype(np.float64).T + bn + Pn[src, 3 * W:4 * W] + Pn[dst, 4 * W:5 * W]
        scale = np.abs(en).astype(np.float64) @ np.abs(Wn).astype(np.float64).T + np.abs(bn) + np.abs(Pn[src, 3 * W:4 * W]) + np.abs(Pn[dst, 4 * W:5 * W])
        outs = []
        for mode in MODES:
            _lib.set_matmul_mode(mode)
            out = torch.empty(E, W, device=dev)
            part = torch.zeros(4096 * 2 * W, dtype=torch.float64, device=dev)
            nb = C.c_int(0)
            if mode == "f32":
                engine.gemm(engine.NT, e, W3, out, bias=b3)
                engine._call("gnm_edge_t_stats_fwd", E, W, P_(out), P_(Pt), P_(s_), P_(d_), P_(part), C.byref(nb), engine._stream())
            else:
                need = lib.gnm_rowtile_workspace_bytes(5 * W)
                ws = engine.scratch(dev).ws(need)
                engine._call("gnm_edge_t_fused_fwd", E, W, P_(e), P_(W3), P_(b3), P_(Pt), P_(s_), P_(d_), P_(out), P_(part), C.byref(nb), P_(ws), need,
                             engine._stream())
            torch.cuda.synchronize()
            outs.append((mode, out.cpu().numpy()))
        lines.append(report(name, outs, ref, scale))
    # ---- fused gt + NN at H = 256
    lines.append("gnm_edge_bwd_gt_nn[256]   ge_in = ge + gt W3, gt = c (gu - m1 - that m2)   [E,256] x [256,256]   (f32: gnm_edge_bwd_gt + gemm NN)")
    for name, gscale in (("normal", lambda: 1.0), ("gradients of 1e-7", lambda: 1e-7),
                         ("rows over 13 decades", lambda: np.exp(rng.uniform(-25, 5, (E, 1))))):
        gen = (rng.standard_normal((E, W)) * gscale()).astype(np.float32)
        tn = rng.standard_normal((E, W)).astype(np.float32)
        Wn = weights(rng, W, W, 1)
        gam = (1.0 + 0.1 * rng.standard_normal(W)).astype(np.float32)
        mu, rstd = (0.1 * rng.standard_normal(W)).astype(np.float32), (1.0 + 0.1 * rng.random(W)).astype(np.float32)
        shift = (0.1 * rng.standard_normal(W)).astype(np.float32)
        stat = np.stack([mu, rstd, gam * rstd, shift - mu * gam * rstd]).astype(np.float32)
        m1 = (np.abs(gen).mean() * 0.01 * rng.standard_normal(W)).astype(np.float32)
        m2 = (np.abs(gen).mean() * 0.01 * rng.standard_normal(W)).astype(np.float32)
        bst = np.stack([m1, m2]).astype(np.float32)
        ge, t, W3, g_, st_, bs_ = (torch.from_numpy(a).to(dev) for a in (gen, tn, Wn, gam, stat, bst))
        t64, ge64 = tn.astype(np.float64), gen.astype(np.float64)
        u32 = np.float32(tn) * stat[2] + stat[3]                 # the kernels decide the branch on the fp32 fma; ties are measure zero here
        gu = ge64 * (u32 > 0)
        that = (t64 - mu) * rstd.astype(np.float64)
        gt64 = (gam.astype(np.float64) * rstd) * (gu - m1 - that * m2)
        ref = ge64 + gt64 @ Wn.astype(np.float64)
        scale = np.abs(ge64) + np.abs(gt64) @ np.abs(Wn.astype(np.float64))
        outs = []
        for mode in MODES:
            _lib.set_matmul_mode(mode)
            gt = torch.empty(E, W, device=dev)
            out = torch.empty(E, W, device=dev)
            if mode == "f32":
                engine._call("gnm_edge_bwd_gt", E, W, P_(ge), P_(t), P_(st_), P_(bs_), P_(g_), P_(gt), engine._stream())
                engine.gemm(engine.NN, gt, W3, out, resid=ge)
            else:
                need = lib.gnm_rowtile_workspace_bytes(5 * W)
                ws = engine.scratch(dev).ws(need)
                engine._call("gnm_edge_bwd_gt_nn", E, W, P_(ge), P_(t), P_(st_), P_(bs_), P_(g_), P_(W3), P_(gt), P_(out), P_(ws), need, engine._stream())
            torch.cuda.synchronize()
            outs.append((mode, out.cpu().numpy()))
        lines.append(report(name, outs, ref, scale))


if __name__ == "__main__":
    main()
