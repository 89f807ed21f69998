#!/usr/bin/env python3
"""A/B of the node-projection kernels (gnm_node_proj_fwd): rowtile_nt_k<MmB3, false, 1> against node_proj16_k (eight waves on
16-column blocks, gnm_debug_set_variant("proj16", 1)) at the metric's N = 1.5 M: agreement with each other and with fp64 on a
row sample, HIP-event time per launch.  Run on the GPU box."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_assembly_amd import _lib, engine    # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    H = 128
    rng = np.random.default_rng(0)
    for N in (1_500_000, 220_000, 4097):
        h = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(dev)
        W5 = torch.from_numpy((rng.standard_normal((5 * H, H)) / 11).astype(np.float32)).to(dev)
        b5 = torch.from_numpy(rng.standard_normal(5 * H).astype(np.float32)).to(dev)
        need = lib.gnm_rowtile_workspace_bytes(5 * H)
        ws = engine.scratch(dev).ws(need)
        out = {}
        for v in (0, 1):
            assert lib.gnm_debug_set_variant(b"proj16", v) == 0
            P = torch.empty(N, 5 * H, device=dev)
            call = lambda: engine._call("gnm_node_proj_fwd", N, H, 5 * H, engine._ptr(h), engine._ptr(W5), engine._ptr(b5), engine._ptr(P),   # noqa: E731
                                        engine._ptr(ws), need, engine._stream())
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            out[v] = (P, e0.elapsed_time(e1) / 20)
        lib.gnm_debug_set_variant(b"proj16", 0)
        idx = torch.from_numpy(rng.choice(N, min(N, 4096), replace=False)).to(dev)
        ref = h[idx].double() @ W5.double().t() + b5.double()
        r0 = float((out[0][0][idx].double() - ref).norm() / ref.norm())
        r1 = float((out[1][0][idx].double() - ref).norm() / ref.norm())
        d = float((out[0][0] - out[1][0]).abs().max())
        print(f"N={N}: rowtile_nt_k {out[0][1]:.3f} ms (rel-L2 vs fp64 {r0:.2e}), node_proj16_k {out[1][1]:.3f} ms ({r1:.2e}); "
              f"max |difference| {d:.2e}; finite {bool(torch.isfinite(out[1][0]).all())}")


if __name__ == "__main__":
    main()
