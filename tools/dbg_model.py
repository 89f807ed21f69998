import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import sd_to_torch, rel_l2
import gnnome_assembly_amd as G
from gnnome_assembly_amd import synth
from oracle import gatedgcn_oracle as orc
H, L, bn = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1"
dev = torch.device("cuda:0")
src, dst, n = synth.make_graph(700, seed=H + L, permute_edge_ids=True)
inp = synth.make_inputs(src, dst, n, seed=H)
sd = synth.synth_state_dict(H, L, seed=L)
model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, bn, 16); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.to(dev)
g = G.AssemblyGraph(src, dst, n).to(dev)
s = model(g, None, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev))
loss = G.BCEWithLogitsLoss(float(inp["pos_weight"]))(s.squeeze(-1), torch.from_numpy(inp["y"]).to(dev)); loss.backward()
out = {}
for dt in (torch.float64, torch.float32):
    p = sd_to_torch(sd, dt, requires_grad=True)
    so = orc.model_forward(p, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).to(dt), torch.from_numpy(inp["pe"]).to(dt), bn)
    orc.bce_loss(so, torch.from_numpy(inp["y"]).to(dt), float(inp["pos_weight"])).backward()
    out[dt] = {k: v.grad.double().numpy() for k, v in p.items()}
print("logits rel", rel_l2(s.detach().cpu().numpy(), so.detach().numpy()))
for k, prm in model.named_parameters():
    got = prm.grad.detach().cpu().double().numpy()
    print(f"{k:28s} ours={rel_l2(got, out[torch.float64][k]):.2e} ref32={rel_l2(out[torch.float32][k], out[torch.float64][k]):.2e} norm={np.linalg.norm(out[torch.float64][k]):.2e}")
