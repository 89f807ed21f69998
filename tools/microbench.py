#!/usr/bin/env python3
"""One GatedGCN layer fwd+bwd at chr19 scale through the C ABI, repeated; prints per-op HIP-event
times.  Meant to be run bare or under `rocprofv3 --pmc ...` to study single kernels."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=750000)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--unfused", action="store_true")
    ap.add_argument("--matmul", default="f32", choices=["f32", "bf16x3", "f16x2"])
    a = ap.parse_args()
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth, engine
    engine.set_default(FUSED=not a.unfused)
    G._lib.set_matmul_mode(a.matmul)
    dev = torch.device("cuda:0")
    H = a.hidden
    src, dst, n = synth.make_graph(a.reads, 0)
    E = src.size
    g = G.AssemblyGraph(src, dst, n, node_order="keep").to(dev)     # engine-level calls: node rows as they are
    idx = g.index()
    sd = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_state_dict(H, 1, 0).items()}
    prm = engine.layer_params(sd, 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    h = torch.randn(n, H, device=dev, generator=gen)
    e = torch.randn(E, H, device=dev, generator=gen)
    gh = torch.randn(n, H, device=dev, generator=gen) * 1e-3
    ge0 = torch.randn(E, H, device=dev, generator=gen) * 1e-3
    for it in range(a.iters + 1):
        if it == 1:
            engine.profile_ops(True)
        h1, e1, s = engine.layer_forward(idx, n, E, H, prm, h, e, True)
        ge = ge0.clone()
        engine.layer_backward(idx, n, E, H, prm, s, gh, ge)
    ops = engine.profile_ops(False)
    # wall time of the same layer without per-op events (the two-stream mode only runs un-profiled)
    import time
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(a.iters):
            h1, e1, s = engine.layer_forward(idx, n, E, H, prm, h, e, True)
            ge = ge0.clone()
            engine.layer_backward(idx, n, E, H, prm, s, gh, ge)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters * 1e3
    wall_msg = f"layer fwd+bwd wall {wall:.2f} ms (incl. one [E,H] clone)"
    tot = 0.0
    for k, (c, t) in sorted(ops.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:34s} calls={c:3d} avg_ms={t / c:8.3f}")
        tot += t / a.iters
    print(f"layer fwd+bwd total {tot:.2f} ms  (E={E}, N={n}, H={H})")
    print(wall_msg)


if __name__ == "__main__":
    main()
