#!/usr/bin/env python3
"""One pass of ClusterGCN mini-batches over a chr19-scale graph (train.py:288-343 counterpart): time per
batch of sub-graph construction (mask, compaction, device index) and of the training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnome_assembly_amd as G
from gnnome_assembly_amd import synth, cluster, dp

R = int(sys.argv[1]) if len(sys.argv) > 1 else 750000
dev = torch.device("cuda:0")
src, dst, n = synth.make_graph(R, seed=0)
inp = synth.make_inputs(src, dst, n, seed=0)
g = G.AssemblyGraph(src, dst, n).to(dev)
g.ndata["pe"] = torch.from_numpy(inp["pe"]).to(dev)
g.edata["e"] = torch.from_numpy(inp["e"]).to(dev)
g.edata["y"] = torch.from_numpy(inp["y"]).to(dev)
t0 = time.perf_counter(); part = cluster.partition_graph(g, 500); t_part = time.perf_counter() - t0
model = G.GraphGatedGCNModel(1, 2, 128, 16, 8, 64, True, 16)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(128, 8, 0, randomize_norm=False).items()})
model.to(dev)
crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
flat = dp.FlatGradients(model.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
for ep in range(2):
    t_sub = t_step = 0.0; ne = 0; nb = 0
    it = iter(cluster.ClusterBatchLoader(g, part, 50, shuffle=True))
    while True:
        torch.cuda.synchronize(); a = time.perf_counter()
        try:
            sub = next(it)
        except StopIteration:
            break
        sub.index()
        torch.cuda.synchronize(); b = time.perf_counter()
        flat.zero_()
        loss = crit(model(sub, None, sub.edata["e"], sub.ndata["pe"]).squeeze(-1), sub.edata["y"])
        loss.backward(); opt.step()
        torch.cuda.synchronize(); c = time.perf_counter()
        t_sub += b - a; t_step += c - b; ne += sub.num_edges(); nb += 1
    print(f"epoch {ep}: {nb} batches, {ne} of {src.size} edges kept ({100*ne/src.size:.1f} %), sub-graph build "
          f"{1e3*t_sub/nb:.1f} ms/batch, step {1e3*t_step/nb:.1f} ms/batch -> {ne/(t_sub+t_step)/1e6:.1f} M edges/s "
          f"(partition once: {t_part:.1f} s)")
