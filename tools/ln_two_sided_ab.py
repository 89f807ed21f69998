#!/usr/bin/env python3
"""LayerNorm model (batch_norm=False) at the metric's graph: training step with the two-sided forward sweep (gnm_ln_edge_gate2_fwd,
default) against the separate passes (gnm_ln_edge_gate_fwd + gnm_node_agg_src_fwd), alternating.  gpurun_out/ln_two_sided.txt"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp, engine, synth
    dev = torch.device("cuda:0")
    R, H, L = int(os.environ.get("READS", "750000")), 128, 8
    src, dst, n = synth.make_graph(R, seed=0)
    inp = synth.make_inputs(src, dst, n, seed=0)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    g.index()
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, False, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
    model.to(dev)
    model.flatten_parameters()
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    flat = dp.FlatGradients(model.parameters(), direct_write=True)
    opt = dp.make_adam(model.parameters(), 1e-3)

    def step():
        flat.zero_()
        loss = crit(model(g, None, e, pe).squeeze(-1), y)
        loss.backward()
        opt.step()

    lines = [f"# LayerNorm model, R={R} (E={src.size}), H={H}, L={L}, matmul={G._lib.get_matmul_mode()}: ms per training step, 10 steps each, alternating"]
    for rep in range(3):
        for two in (True, False):
            with engine.options(TWO_SIDED_FWD=two):
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                lines.append(f"{'two-sided forward sweep' if two else 'separate passes       '}  {(time.perf_counter() - t0) * 100:.2f}")
    # per-op times of one (serialised, event-bracketed) step under the default switches
    engine.profile_ops(True)
    step()
    ops = engine.profile_ops(False)
    lines.append("# per-op times of one step (ms; calls):")
    for k, (c, t) in sorted(ops.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"#   {t:8.2f}  x{c:3d}  {k}")
    lines.append(f"#   {sum(t for _, t in ops.values()):8.2f}  total")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "ln_two_sided.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
