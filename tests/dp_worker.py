"""One rank of the data-parallel HIP-path check (tests/test_gpu_dp.py launches two of these on ONE GPU with
the gloo backend; on an 8-GPU node the same code runs one rank per device over RCCL).

Each rank: the real GraphGatedGCNModel (HIP kernels through libgnm.so) forward + BCE + backward on ITS graph
(seed = rank), the single flat-gradient all-reduce of dp.FlatGradients, one Adam step.  It writes its own
(pre-exchange) gradient, the exchanged flat gradient and the updated parameters for the parent to compare.
Usage: RANK / WORLD_SIZE / MASTER_* in the env;  python dp_worker.py <outdir> <H> <L> <reads>."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out, H, L, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp, synth
    rank, world = dp.init_process_group(os.environ.get("GNM_BENCH_BACKEND", "gloo"))
    dev = torch.device("cuda", int(os.environ.get("GNM_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    src, dst, n = synth.make_graph(R, seed=rank, permute_edge_ids=True)
    inp = synth.make_inputs(src, dst, n, seed=rank)
    sd = synth.synth_state_dict(H, L, seed=0)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(dev)
    graph = G.AssemblyGraph(src, dst, n).to(dev)
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    # dataset-level pos_weight (train.py:181): the same constant on every rank
    pw = torch.tensor([float(inp["pos_weight"])], dtype=torch.float64, device=dev)
    dist.all_reduce(pw)
    crit = G.BCEWithLogitsLoss(float(pw.item()) / world)
    model.flatten_parameters()
    flat = dp.FlatGradients(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    flat.zero_()
    loss = crit(model(graph, None, e, pe).squeeze(-1), y)
    loss.backward()
    torch.cuda.synchronize()
    own = flat.grads.detach().cpu().numpy().copy()
    flat.all_reduce_mean()
    torch.cuda.synchronize()
    red = flat.grads.detach().cpu().numpy().copy()
    opt.step()
    torch.cuda.synchronize()
    w = torch.cat([p.detach().reshape(-1) for p in flat.params]).cpu().numpy()
    names = [k for k, p in model.named_parameters()]
    order = [names[[id(q) for _, q in model.named_parameters()].index(id(p))] for p in flat.params]
    np.savez(os.path.join(out, f"rank{rank}.npz"), own=own, reduced=red, w=w, loss=float(loss.item()),
             pos_weight=float(pw.item()) / world, order=np.array(order))
    print(f"[dp_worker rank {rank}/{world}] device {torch.cuda.get_device_name(dev)} backend {dist.get_backend()} "
          f"E={src.size} loss={loss.item():.6f} |g_own|={np.linalg.norm(own):.4e} |g_mean|={np.linalg.norm(red):.4e}",
          flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
