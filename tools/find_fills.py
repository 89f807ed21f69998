#!/usr/bin/env python3
"""Where the aten fill / copy launches of one training step come from (torch profiler with stacks): RR=20000 python tools/find_fills.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnome_assembly_amd as G
from gnnome_assembly_amd import synth, engine, dp
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
R, H, L = int(os.environ.get('RR', 20000)), 128, 8
src, dst, n = synth.make_graph(R, seed=0)
inp = synth.make_inputs(src, dst, n, seed=0)
graph = G.AssemblyGraph(src, dst, n).to(dev); graph.index()
model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
model.to(dev); model.flatten_parameters()
e = torch.from_numpy(inp["e"]).to(dev); pe = torch.from_numpy(inp["pe"]).to(dev); y = torch.from_numpy(inp["y"]).to(dev)
crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
flat = dp.FlatGradients(model.parameters(), direct_write=True)
opt = dp.make_adam(model.parameters(), 1e-3)
def step():
    flat.zero_(); loss = crit(model(graph, None, e, pe).squeeze(-1), y); loss.backward(); flat.all_reduce_mean(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy"):
        st = [f for f in ev.stack if "gnnome_assembly_amd" in f or "bench" in f or "optim" in f][:2]
        c[(ev.name, tuple(st))] += 1
for (name, st), k in c.most_common(25):
    print(k, name, st)

kc = Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith('CUDA'):
        kc[ev.name[:60]] += 1
print([ (k,v) for k,v in kc.most_common(40) if 'Fill' in k or 'copy' in k.lower() or 'Memcpy' in k or 'Memset' in k])
