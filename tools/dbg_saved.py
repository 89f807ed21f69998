import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import sd_to_torch, rel_l2
import gnnome_assembly_amd as G
from gnnome_assembly_amd import synth, engine
from oracle import gatedgcn_oracle as orc
H, L = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
src, dst, n = synth.make_graph(700, seed=H + L, permute_edge_ids=True)
inp = synth.make_inputs(src, dst, n, seed=H)
sd = synth.synth_state_dict(H, L, seed=L)
P = {k: v.to(dev) for k, v in sd_to_torch(sd).items()}
g = G.AssemblyGraph(src, dst, n).to(dev); idx = g.index(); perm = idx["perm"].long().cpu()
p64 = sd_to_torch(sd, torch.float64)
with torch.no_grad():
    sc64, lo64, g64, dbg = orc.manual_forward_backward(p64, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).double(), torch.from_numpy(inp["pe"]).double(), torch.from_numpy(inp["y"]).double(), float(inp["pos_weight"]), keep=True)
scores, ms = engine.model_forward(g, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev), P, L, True)
torch.cuda.synchronize()
def snap():
    out = {}
    for i, s in enumerate(ms.layers):
        for k in ("h_in", "e_in", "P", "t", "e_out", "hf", "inv_f", "hb", "inv_b", "z", "stat_e", "stat_h"):
            out[f"L{i}.{k}"] = getattr(s, k).clone()
    out["pred.x"] = ms.pred.x.clone(); out["pred.e"] = ms.pred.e.clone(); out["pred.hid"] = ms.pred.hid.clone(); out["pred.W1sd"] = ms.pred.W1sd.clone()
    return out
before = snap()
for i in range(L):
    d = dbg[i]
    for k, ref in (("h_in", d["h"]), ("e_in", d["e"][perm]), ("P", d["P"]), ("t", d["t"][perm]), ("e_out", d["e_out"][perm]), ("z", d["z"])):
        print(f"fwd L{i}.{k:6s} rel={rel_l2(before[f'L{i}.{k}'].cpu().double().numpy(), ref.numpy()):.2e}")
print("fwd pred.hid", rel_l2(before["pred.hid"].cpu().double().numpy(), dbg["hid"][perm].numpy()))
print("ptrs:", {k: hex(v.data_ptr()) for k, v in list(before.items())[:0]})
loss, gs = engine.bce_with_logits(scores, torch.from_numpy(inp["y"]).to(dev), float(inp["pos_weight"]))
# hold references to the saved tensors so we can re-inspect after backward
layers = list(ms.layers); pred = ms.pred
G_ = engine.model_backward(g, P, L, ms, gs)
torch.cuda.synchronize()
for i, s in enumerate(layers):
    for k in ("h_in", "e_in", "P", "t", "e_out", "hf", "inv_f", "hb", "inv_b", "z", "stat_e", "stat_h"):
        a, b = getattr(s, k), before[f"L{i}.{k}"]
        if not torch.equal(a, b):
            print(f"CHANGED during backward: L{i}.{k}  maxdiff={float((a-b).abs().max()):.3e} n={(a!=b).sum().item()}")
for k in ("x", "e", "W1sd"):
    a, b = getattr(pred, k), before["pred." + k]
    if not torch.equal(a, b): print("CHANGED pred." + k, float((a-b).abs().max()))
for k in ("predictor.W1.weight", "gnn.convs.%d.A_1.weight" % (L-1), "gnn.convs.0.A_1.weight", "linear_pe.weight"):
    print(k, "rel", rel_l2(G_[k].cpu().double().numpy(), g64[k].numpy()))
