"""One rank of BASELINE.json configs[3] in miniature: gnnome_assembly_amd.train.train -- the counterpart of the
reference's full-graph loop (train.py:232-281,379-529) -- under torch.distributed on the HIP path, on a mixed
chr19 / chr20 / chr21 training set (relative sizes of evaluate.py:28-30) sharded over the ranks with
dp.shard_graphs, plus one validation graph.  tests/test_gpu_dp.py launches two of these on ONE GPU with gloo; on an
8-GPU node the same code runs one rank per device over RCCL.
Usage: RANK / WORLD_SIZE / MASTER_* in the env;  python train_dp_worker.py <outdir> <H> <L> <base reads> <epochs>."""
import hashlib
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CHROMS = ("chr19", "chr20", "chr21")


def dataset(base_reads):
    """(train graphs, validation graph) as (reads, seed) pairs: three training graphs at chr19 : chr20 : chr21 sizes."""
    from gnnome_assembly_amd import synth
    train = [(int(round(base_reads * synth.CHR_SCALE[c])), i) for i, c in enumerate(CHROMS)]
    return train, (int(round(base_reads * 0.5)), 100)


def main():
    out, H, L, R, epochs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp, synth, train as T
    rank, world = dp.init_process_group(os.environ.get("GNM_BENCH_BACKEND", "gloo"))
    dev = torch.device("cuda", int(os.environ.get("GNM_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)

    def sample(reads, seed):
        src, dst, n = synth.make_graph(reads, seed=seed, permute_edge_ids=True)
        inp = synth.make_inputs(src, dst, n, seed=seed)
        g = G.AssemblyGraph(src, dst, n).to(dev)
        return T.GraphSample(g, *(torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y")))

    train_set, valid = dataset(R)
    mine = dp.shard_graphs(len(train_set), rank, world, sizes=[r for r, _ in train_set])   # size-sorted, round-robin
    mine_valid = dp.shard_graphs(1, rank, world)                                           # rank 0 holds the one graph
    tr = [sample(*train_set[i]) for i in mine]
    va = [sample(*valid) for _ in mine_valid]
    first = {}
    hashes = []

    def after_exchange(epoch, it, flat):
        if epoch == 0 and it == 0:
            first["grad"] = flat.grads.detach().cpu().numpy().copy()      # in the order of the flat buffer (p._gnm_slot)

    def after_epoch(epoch, model):
        torch.cuda.synchronize()
        sd = model.state_dict()
        h = hashlib.sha256()
        for k in sd:
            h.update(sd[k].detach().cpu().numpy().tobytes())
        hashes.append(h.hexdigest())

    hp = dict(num_epochs=epochs, dim_latent=H, num_gnn_layers=L, lr=1e-3, patience=0, decay=0.5, seed=0)
    workdir = os.path.join(out, f"rank{rank}")
    os.makedirs(workdir, exist_ok=True)
    model, best, hist = T.train(tr, va, out="cfg4", hyperparameters=hp, workdir=workdir, verbose=rank == 0,
                                hooks={"after_exchange": after_exchange, "after_epoch": after_epoch})
    torch.cuda.synchronize()
    names = {id(p): k for k, p in model.named_parameters()}
    flat_order = sorted((p for p in model.parameters()), key=lambda p: p._gnm_slot)
    np.savez(os.path.join(out, f"rank{rank}.npz"), grad0=first["grad"], order=np.array([names[id(p)] for p in flat_order]),
             final=torch.cat([p.detach().reshape(-1) for p in flat_order]).cpu().numpy())
    json.dump({"rank": rank, "world": world, "backend": dist.get_backend(), "shard": mine, "valid_shard": mine_valid,
               "step_graph": hist.step_graph, "step_losses": hist.step_losses, "loss_train": hist.loss_train,
               "loss_valid": hist.loss_valid, "lr": hist.lr, "final_lr": hist.final_lr, "best_epoch": hist.best_epoch,
               "tfpn_train": hist.tfpn_train, "tfpn_valid": hist.tfpn_valid, "epoch_hashes": hashes,
               "files": sorted(os.path.relpath(os.path.join(d, f), workdir) for d, _, fs in os.walk(workdir) for f in fs)},
              open(os.path.join(out, f"rank{rank}.json"), "w"))
    print(f"[train_dp_worker rank {rank}/{world}] backend {dist.get_backend()} shard {mine} "
          f"E={[s.graph.num_edges() for s in tr]} steps/epoch={len(hist.step_graph) // max(epochs, 1)} "
          f"loss_train={hist.loss_train} loss_valid={hist.loss_valid} lr={hist.lr} best_epoch={hist.best_epoch}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
