import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gnnome_assembly_amd import synth, models, AssemblyGraph, _lib
from oracle import gatedgcn_oracle as orc
dev = torch.device("cuda:0")
for seed in (3, 4, 5, 6, 7, 8):
    src, dst, n = synth.make_graph(500, seed=seed)
    inp = synth.make_inputs(src, dst, n, seed=seed)
    H, L = 128, 2
    sd = synth.synth_state_dict(H, L, seed=seed)
    def oracle(dt):
        p = {k: torch.from_numpy(v).to(dt).requires_grad_(True) for k, v in sd.items()}
        s = orc.model_forward(p, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).to(dt), torch.from_numpy(inp["pe"]).to(dt))
        l = orc.bce_loss(s, torch.from_numpy(inp["y"]).to(dt), float(inp["pos_weight"]))
        l.backward()
        return {k: v.grad.double().numpy() for k, v in p.items()}
    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    for mode in ("bf16x3", "f32"):
        _lib.set_matmul_mode(mode)
        model = models.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.to(dev)
        g = AssemblyGraph(src, dst, n).to(dev)
        crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([float(inp["pos_weight"])], device=dev))
        scores = model(g, None, torch.from_numpy(inp["e"]).to(dev), torch.from_numpy(inp["pe"]).to(dev))
        loss = crit(scores.squeeze(-1), torch.from_numpy(inp["y"]).to(dev))
        loss.backward()
        worst = []
        for k, prm in model.named_parameters():
            a = prm.grad.detach().cpu().double().numpy()
            r = np.linalg.norm(a - g64[k]) / max(np.linalg.norm(g64[k]), 1e-12)
            r32 = np.linalg.norm(g32[k] - g64[k]) / max(np.linalg.norm(g64[k]), 1e-12)
            if np.abs(a - g64[k]).max() >= 1e-7: worst.append((r, r32, k))
        worst.sort(reverse=True)
        print(seed, mode, [(f"{r:.1e}", f"{r32:.1e}", k) for r, r32, k in worst[:4]])
