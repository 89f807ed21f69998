#!/usr/bin/env python3
"""Does one whole training step capture into a HIP graph, and what does replay save at small sizes?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnome_assembly_amd as G
from gnnome_assembly_amd import synth, dp

R = int(sys.argv[1]) if len(sys.argv) > 1 else 110000
dev = torch.device("cuda:0")
src, dst, n = synth.make_graph(R, seed=0)
inp = synth.make_inputs(src, dst, n, seed=0)
g = G.AssemblyGraph(src, dst, n).to(dev); g.index()
model = G.GraphGatedGCNModel(1, 2, 128, 16, 8, 64, True, 16)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(128, 8, 0, randomize_norm=False).items()})
model.to(dev)
e = torch.from_numpy(inp["e"]).to(dev); pe = torch.from_numpy(inp["pe"]).to(dev); y = torch.from_numpy(inp["y"]).to(dev)
crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
flat = dp.FlatGradients(model.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
def step():
    flat.zero_()
    loss = crit(model(g, None, e, pe).squeeze(-1), y)
    loss.backward()
    opt.step()
    return loss
def timeit(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print(f"E={src.size} eager {timeit(step):.2f} ms/step")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    loss = step()
print(f"graph replay {timeit(gr.replay):.2f} ms/step  loss={loss.item():.6f}")
