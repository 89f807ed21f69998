#!/usr/bin/env python3
"""Where a mini-batch step's time goes (tools/minibatch_epoch.py's loop, reference settings): the induced sub-graph + its index, the
forward/backward kernels, the optimizer step -- each bracketed by synchronisations -- against the loop as it really runs
(no synchronisation inside).  Prints a table; gpurun_out/minibatch_breakdown.txt."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import cluster, dp, synth
    dev = torch.device("cuda:0")
    H, L = 128, 8
    src, dst, n = synth.make_graph(750000, seed=0)
    inp = synth.make_inputs(src, dst, n, seed=0)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    g.ndata["pe"] = torch.from_numpy(inp["pe"]).to(dev)
    g.edata["e"] = torch.from_numpy(inp["e"]).to(dev)
    g.edata["y"] = torch.from_numpy(inp["y"]).to(dev)
    g.index()
    part = cluster.partition_graph(g, 500, "locality")
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
    model.to(dev)
    model.flatten_parameters()
    flat = dp.FlatGradients(model.parameters(), direct_write=True)
    opt = dp.make_adam(model.parameters(), 1e-3)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    gen = torch.Generator().manual_seed(0)
    sync = torch.cuda.synchronize
    lines = []
    for ep in range(3):
        loader = cluster.ClusterBatchLoader(g, part, 50, shuffle=True, generator=gen)
        acc = {"subgraph": 0.0, "index": 0.0, "forward": 0.0, "backward": 0.0, "optimizer": 0.0}
        host = {"subgraph": 0.0, "index": 0.0, "forward": 0.0, "backward": 0.0, "optimizer": 0.0}
        sync()
        t_ep = time.perf_counter()
        it = iter(loader)
        nsteps = 0
        while True:
            t0 = time.perf_counter()
            try:
                sub = next(it)
            except StopIteration:
                break
            h0 = time.perf_counter(); sync(); t1 = time.perf_counter()
            acc["subgraph"] += t1 - t0; host["subgraph"] += h0 - t0
            sub.index()
            h1 = time.perf_counter(); sync(); t2 = time.perf_counter()
            acc["index"] += t2 - t1; host["index"] += h1 - t1
            flat.zero_()
            s = model(sub, None, sub.edata["e"], sub.ndata["pe"])
            loss = crit(s.squeeze(-1), sub.edata["y"])
            h2 = time.perf_counter(); sync(); t3 = time.perf_counter()
            acc["forward"] += t3 - t2; host["forward"] += h2 - t2
            loss.backward()
            h3 = time.perf_counter(); sync(); t4 = time.perf_counter()
            acc["backward"] += t4 - t3; host["backward"] += h3 - t3
            opt.step()
            h4 = time.perf_counter(); sync(); t5 = time.perf_counter()
            acc["optimizer"] += t5 - t4; host["optimizer"] += h4 - t4
            nsteps += 1
        tot = time.perf_counter() - t_ep
        lines.append(f"epoch {ep} (synchronised after every stage): {tot * 1e3:.1f} ms, {nsteps} steps; per step, ms (host-side part in brackets): "
                     + ", ".join(f"{k} {acc[k] / nsteps * 1e3:.2f} [{host[k] / nsteps * 1e3:.2f}]" for k in acc))
    for ep in range(2):
        loader = cluster.ClusterBatchLoader(g, part, 50, shuffle=True, generator=gen)
        sync()
        t0 = time.perf_counter()
        ne = 0
        for sub in loader:
            flat.zero_()
            s = model(sub, None, sub.edata["e"], sub.ndata["pe"])
            loss = crit(s.squeeze(-1), sub.edata["y"])
            loss.backward()
            opt.step()
            ne += sub.num_edges()
        h = time.perf_counter() - t0
        sync()
        dt = time.perf_counter() - t0
        lines.append(f"free-running epoch {ep}: {dt * 1e3:.1f} ms (host loop done after {h * 1e3:.1f} ms), {ne / dt / 1e6:.2f} M edges/s")
    out = "\n".join(lines)
    print(out)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "minibatch_breakdown.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
