import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gnnome_assembly_amd as G
from gnnome_assembly_amd import synth, engine, dp, _lib
dev = torch.device("cuda:0")
R, H, L = 750000, 128, 8
src, dst, n = synth.make_graph(R, seed=0)
inp = synth.make_inputs(src, dst, n, seed=0)
graph = G.AssemblyGraph(src, dst, n).to(dev); graph.index()
model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
model.to(dev); model.flatten_parameters()
e = torch.from_numpy(inp["e"]).to(dev); pe = torch.from_numpy(inp["pe"]).to(dev); y = torch.from_numpy(inp["y"]).to(dev)
crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
flat = dp.FlatGradients(model.parameters(), direct_write=True)
for _ in range(3):
    flat.zero_(); loss = crit(model(graph, None, e, pe).squeeze(-1), y); loss.backward()
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
NW = 8
buf = (C.c_longlong * (256 * NW * 12))()
print("rc", lib.gnm_debug_chain_timing(buf))
a = np.frombuffer(buf, dtype=np.int64).reshape(256, NW, 12).astype(np.float64)
nt = a[:, :, 10]
per = a[:, :, :10] / np.maximum(nt[:, :, None], 1)
names = ["loop-top(gather issue->)", "phase0 after the rows arrived", "barrier1", "prefetch+MFMA TN+NN", "barrier2", "dst arithmetic", "barrier3", "walk / bn sums (waves 0-2 / 4-7)", "by-source run sums + gather issue", "phase0: wait for the rows"]
print("tiles per WG:", nt[:4, 0])
print("ticks per tile (s_memtime ticks = 100 MHz? reported raw), mean over WGs, per wave:")
for q in (0, 9, 1, 2, 3, 4, 5, 6, 7, 8):
    print(f"{names[q]:32s}", " ".join(f"{per[:, w, q].mean():8.1f}" for w in range(8)))
print("sum", " ".join(f"{per[:, w, :10].sum(-1).mean():8.1f}" for w in range(8)))
