#!/usr/bin/env python3
"""One mini-batch epoch of the ClusterGCN mode on the chr19-scale graph with the reference's own settings
(hyperparameters.py:15-18: 500 METIS parts, 50 clusters per batch -> 10 optimizer steps per graph and epoch;
train.py:288-293, 296-343): partition time, edge-cut fraction, edges/s and peak memory.  Writes one JSON line
(gpurun_out/minibatch.json -> profiles/r04_minibatch.json).  The partitioner is this package's own (cluster.py; METIS
lives inside DGL): the cut it finds is reported, parity with METIS's cut is not claimed."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=750000)
    ap.add_argument("--parts", type=int, default=500)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--shuffle-nodes", action="store_true")
    ap.add_argument("--method", default="locality")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)     # --hidden 256 --layers 16: the reference's default model (hyperparameters.py:8,13)
    a = ap.parse_args()
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import cluster, dp, synth
    dev = torch.device("cuda:0")
    H, L = a.hidden, a.layers
    src, dst, n = synth.make_graph(a.reads, seed=0)
    inp = synth.make_inputs(src, dst, n, seed=0)
    pe_np = inp["pe"]
    if a.shuffle_nodes:
        p = np.random.default_rng(4242).permutation(n).astype(np.int32)
        src, dst = p[src], p[dst]
        pe_np = np.empty_like(inp["pe"])
        pe_np[p] = inp["pe"]
    E = int(src.size)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    g.ndata["pe"] = torch.from_numpy(pe_np).to(dev)         # [N,18]: in_deg | out_deg | pe of the FULL graph (train.py:301-305)
    g.edata["e"] = torch.from_numpy(inp["e"]).to(dev)
    g.edata["y"] = torch.from_numpy(inp["y"]).to(dev)
    t0 = time.perf_counter()
    g.index()
    t_index = time.perf_counter() - t0
    t0 = time.perf_counter()
    part = cluster.partition_graph(g, a.parts, a.method)
    t_part = time.perf_counter() - t0
    cut = cluster.edge_cut(g, part)
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
    model.to(dev)
    model.flatten_parameters()
    flat = dp.FlatGradients(model.parameters(), direct_write=True)
    opt = dp.make_adam(model.parameters(), 1e-3)
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    gen = torch.Generator().manual_seed(0)
    epochs = []
    torch.cuda.reset_peak_memory_stats()
    for ep in range(a.epochs):
        loader = cluster.ClusterBatchLoader(g, part, a.batch, shuffle=True, generator=gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seen_e = seen_n = 0
        losses = []
        t_sub = 0.0
        for sub in loader:                                   # induced sub-graph, features sliced on the device
            flat.zero_()
            s = model(sub, None, sub.edata["e"], sub.ndata["pe"])
            loss = crit(s.squeeze(-1), sub.edata["y"])
            loss.backward()
            opt.step()
            seen_e += sub.num_edges()
            seen_n += sub.num_nodes()
            losses.append(loss)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        epochs.append({"seconds": round(dt, 4), "steps": len(loader), "edges_in_batches": seen_e, "nodes_in_batches": seen_n,
                       "edges_per_s": seen_e / dt, "mean_loss": float(torch.stack(losses).mean())})
    res = {"what": "one graph, ClusterGCN mini-batch epochs (train.py:288-343 counterpart), reference settings",
           "reads": a.reads, "nodes": n, "edges": E, "hidden": H, "layers": L, "num_parts": a.parts, "clusters_per_batch": a.batch,
           "partition_method": a.method, "node_ids": "shuffled" if a.shuffle_nodes else "position-sorted",
           "index_seconds": round(t_index, 3), "partition_seconds": round(t_part, 3), "edge_cut": cut,
           "edge_cut_fraction": cut / E, "edges_kept_per_epoch_fraction": epochs[-1]["edges_in_batches"] / E,
           "epochs": epochs, "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "steady_state_edges_per_s": float(np.median([ep["edges_per_s"] for ep in epochs[2:]])) if len(epochs) > 2 else None,
           "note": "epochs 0 and 1 include the one-off costs (first sub-graph indices, allocator growth, optimizer state); the induced sub-graphs are born "
                   "on the device (graph.tensor_index) and run the separate-pass schedule (no sweep plan for device-born graphs)"}
    print(json.dumps(res))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    tag = ("_shuffled" if a.shuffle_nodes else "") + (f"_h{H}l{L}" if (H, L) != (128, 8) else "")
    with open(os.path.join(REPO, "gpurun_out", "minibatch" + tag + ".json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
