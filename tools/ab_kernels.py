#!/usr/bin/env python3
"""Within-process A/B of kernel generations behind one C-ABI entry point (gnm_debug_set_variant), at chr19
scale: interleaved rounds, HIP-event times (median / min), and the results of the two variants compared with
each other and with an fp64 evaluation on a sub-sample.  Run on the GPU box:  python tools/ab_kernels.py"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timed(fn, rounds):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=7_540_278)
    ap.add_argument("--nodes", type=int, default=1_500_000)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--which", default="edge_bwd,tn")
    ap.add_argument("--eb-variants", default="0,1,2")
    a = ap.parse_args()
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import engine, _lib
    lib = _lib.load()
    _lib.set_matmul_mode("bf16x3")
    dev = torch.device("cuda:0")
    H = 128
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
    sc, st, ptr = engine.scratch(dev), engine._stream(), engine._ptr
    which = a.which.split(",")
    if "edge_bwd" in which:
        E = a.edges
        ge0, t, e_in = rnd(E, H) * 1e-3, rnd(E, H), rnd(E, H)
        W3 = rnd(H, H) / H ** 0.5
        gamma = torch.rand(H, device=dev, generator=gen) + 0.5
        mean, rstd = t.mean(0), 1.0 / (t.var(0, unbiased=False) + 1e-5).sqrt()
        beta = rnd(H) * 0.1
        stat = torch.stack([mean, rstd, gamma * rstd, beta - mean * gamma * rstd]).contiguous()
        bstat = torch.stack([rnd(H) * 1e-4, rnd(H) * 1e-4]).contiguous()
        need = lib.gnm_edge_bwd_fused_workspace_bytes()
        ws = sc.ws(need)
        out = {}
        EV = [int(x) for x in a.eb_variants.split(",")]
        for v in EV:
            lib.gnm_debug_set_variant(b"edge_bwd", v)
            ge_out = torch.empty_like(ge0)
            gW3, gb3 = torch.empty(H, H, device=dev), torch.empty(H, device=dev)

            def run():
                engine._call("gnm_edge_bwd_fused", E, H, ptr(ge0), ptr(ge_out), ptr(t), ptr(e_in), ptr(stat), ptr(bstat),
                             ptr(gamma), ptr(W3), ptr(gW3), ptr(gb3), ptr(sc.partials), ptr(ws), need, st)
            run()
            torch.cuda.synchronize()
            out[v] = (ge_out.clone(), gW3.clone(), gb3.clone(), run)
        res = {v: [] for v in EV}
        for _ in range(a.rounds):
            for v in EV:
                lib.gnm_debug_set_variant(b"edge_bwd", v)
                res[v].append(timed(out[v][3], 1)[0])
        lib.gnm_debug_set_variant(b"edge_bwd", 1)
        for v in EV:
            print(f"edge_bwd_fused variant {v}: median {np.median(res[v]):.3f} ms  min {np.min(res[v]):.3f} ms  "
                  f"({4 * E * H * 4 / np.median(res[v]) / 1e9:.2f} TB/s algorithmic)")
        for v in EV[1:]:
            print(f"  variant {v} vs {EV[0]}: " + "  ".join(f"{name} {rel(out[v][i], out[EV[0]][i]):.3e}" for name, i in (("ge_out", 0), ("gW3", 1), ("gb3", 2))))
        # fp64 check on the first 200k rows (ge_out) -- and of gW3 / gb3 over everything in fp64 chunks
        n = min(E, 200_000)
        gu = torch.where(t[:n] * stat[2] + stat[3] > 0, ge0[:n], torch.zeros_like(ge0[:n])).double()
        gt = (gamma * stat[1]).double() * (gu - bstat[0].double() - ((t[:n].double() - mean.double()) * rstd.double()) * bstat[1].double())
        want = ge0[:n].double() + gt @ W3.double()
        for v in EV:
            print(f"  variant {v}: ge_out vs fp64 (first {n} rows) rel_l2 = {rel(out[v][0][:n], want):.3e}")
        gW = torch.zeros(H, H, dtype=torch.float64, device=dev)
        gb = torch.zeros(H, dtype=torch.float64, device=dev)
        for s0 in range(0, E, 1 << 20):
            sl = slice(s0, min(E, s0 + (1 << 20)))
            gu = torch.where(t[sl] * stat[2] + stat[3] > 0, ge0[sl], torch.zeros_like(ge0[sl])).double()
            gt = (gamma * stat[1]).double() * (gu - bstat[0].double() - ((t[sl].double() - mean.double()) * rstd.double()) * bstat[1].double())
            gW += gt.T @ e_in[sl].double()
            gb += gt.sum(0)
        for v in EV:
            print(f"  variant {v}: gW3 vs fp64 rel_l2 = {rel(out[v][1], gW):.3e}   gb3 rel_l2 = {rel(out[v][2], gb):.3e}")
        del ge0, t, e_in, out
        torch.cuda.empty_cache()
    if "tn" in which:
        N = a.nodes
        gP, h = rnd(N, 5 * H) * 1e-3, rnd(N, H)
        need = lib.gnm_node_proj_bwd_workspace_bytes(5 * H)
        ws = sc.ws(need)
        out = {}
        for v in (0, 1):
            lib.gnm_debug_set_variant(b"tn", v)
            gW, gb = torch.empty(5 * H, H, device=dev), torch.empty(5 * H, device=dev)

            def run():
                engine._call("gnm_node_proj_bwd_tn", N, H, 5 * H, ptr(gP), ptr(h), ptr(gW), ptr(gb), ptr(sc.partials), ptr(ws), need, 0, st)
            run()
            torch.cuda.synchronize()
            out[v] = (gW.clone(), gb.clone(), run)
        res = {v: [] for v in (0, 1)}
        for _ in range(a.rounds):
            for v in (0, 1):
                lib.gnm_debug_set_variant(b"tn", v)
                res[v].append(timed(out[v][2], 1)[0])
        lib.gnm_debug_set_variant(b"tn", 1)
        want = torch.zeros(5 * H, H, dtype=torch.float64, device=dev)
        for s0 in range(0, N, 1 << 19):
            sl = slice(s0, min(N, s0 + (1 << 19)))
            want += gP[sl].double().T @ h[sl].double()
        wb = gP.double().sum(0)
        for v in (0, 1):
            print(f"node_proj_bwd_tn variant {v}: median {np.median(res[v]):.3f} ms  min {np.min(res[v]):.3f} ms  "
                  f"({6 * N * H * 4 / np.median(res[v]) / 1e9:.2f} TB/s algorithmic);  gW5 vs fp64 {rel(out[v][0], want):.3e}  "
                  f"gb5 vs fp64 {rel(out[v][1], wb):.3e}")


if __name__ == "__main__":
    main()
