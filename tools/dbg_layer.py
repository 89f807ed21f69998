import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from helpers import sd_to_torch, rel_l2
from gnnome_assembly_amd import AssemblyGraph, engine, synth
from oracle import gatedgcn_oracle as orc
H, L = int(sys.argv[1]), 1
dev = torch.device("cuda:0")
src, dst, n = synth.make_graph(700, seed=3, permute_edge_ids=True)
inp = synth.make_inputs(src, dst, n, seed=3)
sd = synth.synth_state_dict(H, L, seed=1)
p64 = sd_to_torch(sd, torch.float64)
with torch.no_grad():
    _, _, g64, dbg = orc.manual_forward_backward(p64, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).double(), torch.from_numpy(inp["pe"]).double(), torch.from_numpy(inp["y"]).double(), float(inp["pos_weight"]), keep=True)
graph = AssemblyGraph(src, dst, n).to(dev); idx = graph.index(); perm = idx["perm"].long().cpu(); E = src.size
d = dbg[0]
P32 = {k: v.to(dev) for k, v in sd_to_torch(sd).items()}
prm = engine.layer_params(P32, 0)
f = lambda t: t.float().contiguous().to(dev)
h_out, e_out, s = engine.layer_forward(idx, n, E, H, prm, f(d["h"]), f(d["e"][perm]), True)
gh_in, ge_in, g = engine.layer_backward(idx, n, E, H, prm, s, f(d["gh_out"]), f(d["ge_out"][perm]))
torch.cuda.synchronize()
def c(name, a, b): print(f"{name:12s} rel={rel_l2(a.detach().cpu().double().numpy(), b.double().numpy()):.3e}")
c("e_out", e_out, d["e_out"][perm]); c("h_out", h_out, d["h_out"])
c("gh_in", gh_in, d["gh_in"]); c("ge_in", ge_in, d["ge_in"][perm])
gW5 = torch.cat([g64[f"gnn.convs.0.{k}.weight"] for k in engine.LIN5], 0); gb5 = torch.cat([g64[f"gnn.convs.0.{k}.bias"] for k in engine.LIN5], 0)
for j,k in enumerate(engine.LIN5): c("gW5."+k, g["W5"][j*H:(j+1)*H], gW5[j*H:(j+1)*H])
c("gb5[A2,A3]", g["b5"][H:3*H], gb5[H:3*H]); c("gW3", g["W3"], g64["gnn.convs.0.B_3.weight"])
c("ggamma_e", g["gamma_e"], g64["gnn.convs.0.bn_e.weight"]); c("gbeta_e", g["beta_e"], g64["gnn.convs.0.bn_e.bias"])
c("ggamma_h", g["gamma_h"], g64["gnn.convs.0.bn_h.weight"]); c("gbeta_h", g["beta_h"], g64["gnn.convs.0.bn_h.bias"])
