#!/usr/bin/env python3
"""Golden vectors for the training-harness row (SURVEY.md section 8f rank 2 / 8c "pins for harness rows"):
the reference's full-graph training loop run on the reference's own model.

Run in the build container only:      python tests/golden/make_golden_train.py

The reference's train.train cannot be imported (wandb, Bio, dgl.dataloading, and `verbose=True` at
train.py:212 is a TypeError on this torch); what IS imported unmodified is everything that decides the
numbers: `models.GraphGatedGCNModel` (+ `layers`), `utils.set_seed`, `utils.calculate_tfpn`,
`utils.calculate_metrics`, and real torch (Adam, BCEWithLogitsLoss, ReduceLROnPlateau).  The loop below is the
statement sequence of train.py:181 (pos_to_neg_ratio), :195-212 (model, Adam, pos_weight, scheduler),
:237-258 (shuffle, per-graph step), :346-353 (epoch means), :385-411 (validation under no_grad / eval),
:525-529 (best model, scheduler.step(val loss)), with the DGL graph replaced by the test-only stand-in.
Only data is written: inputs are regenerated from seeds by gnnome_assembly_amd.synth, the fixture holds the
hyperparameters and the expected per-step / per-epoch numbers in fp32 and fp64.
"""
import copy
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "dgl_standin"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import dgl  # noqa: E402  (the stand-in)
import models  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference)
from torch.optim.lr_scheduler import ReduceLROnPlateau  # noqa: E402
from gnnome_assembly_amd import synth  # noqa: E402

torch.set_num_threads(4)

HP = dict(seed=0, lr=2e-2, num_epochs=4, dim_latent=128, node_features=1, edge_features=2, hidden_edge_features=16,
          hidden_edge_scores=64, num_gnn_layers=2, nb_pos_enc=16, batch_size_train=1, batch_size_eval=1, patience=0,
          decay=0.5, batch_norm=True)
TRAIN = [dict(reads=500, seed=0, permute=False), dict(reads=600, seed=1, permute=True)]
VALID = [dict(reads=400, seed=2, permute=False)]


def load(spec, dtype):
    src, dst, n = synth.make_graph(spec["reads"], spec["seed"], permute_edge_ids=spec["permute"])
    inp = synth.make_inputs(src, dst, n, spec["seed"])
    g = dgl.graph((src, dst), num_nodes=n)
    t = lambda a: torch.from_numpy(a).to(dtype)  # noqa: E731
    return dict(g=g, x=t(inp["x"]), e=t(inp["e"]), pe=t(inp["pe"]), y=t(inp["y"]))


def run(dtype):
    hp = HP
    ref_utils.set_seed(hp["seed"])                                                        # train.py:158
    train, valid = [load(s, dtype) for s in TRAIN], [load(s, dtype) for s in VALID]
    ratio = sum(((d["y"] == 1).sum() / (d["y"] == 0).sum()).item() for d in train) / len(train)   # train.py:181
    model = models.GraphGatedGCNModel(hp["node_features"], hp["edge_features"], hp["dim_latent"], hp["hidden_edge_features"],
                                      hp["num_gnn_layers"], hp["hidden_edge_scores"], hp["batch_norm"], hp["nb_pos_enc"])
    init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}         # fp32 init under seed 0
    model = model.to(dtype)
    optimizer = torch.optim.Adam(model.parameters(), lr=hp["lr"])                         # train.py:209
    criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([1 / ratio], dtype=dtype))    # :210-211
    scheduler = ReduceLROnPlateau(optimizer, mode="min", factor=hp["decay"], patience=hp["patience"])   # :212
    order = list(range(len(train)))
    step_losses, step_graph, tfpn_train, tfpn_valid = [], [], [], []
    loss_train, loss_valid, lrs, best_epoch = [], [], [], -1
    for epoch in range(hp["num_epochs"]):
        random.shuffle(order)                                                             # :238
        ep, counts = [], np.zeros(4, dtype=np.int64)
        for gi in order:
            d = train[gi]
            model.train()
            pred = model(d["g"], d["x"], d["e"], d["pe"]).squeeze(-1)                     # :252-253
            loss = criterion(pred, d["y"])
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()                                                              # :256-258
            ep.append(loss.item())
            step_graph.append(gi)
            counts += np.array(ref_utils.calculate_tfpn(pred, d["y"]), dtype=np.int64)    # :260
        step_losses += ep
        tfpn_train.append(counts)
        loss_train.append(float(np.mean(ep)))                                             # :347
        lrs.append(optimizer.param_groups[0]["lr"])                                       # :353
        vl, vcounts = [], np.zeros(4, dtype=np.int64)
        with torch.no_grad():                                                             # :385-411
            model.eval()
            for d in valid:
                pred = model(d["g"], d["x"], d["e"], d["pe"]).squeeze(-1)
                vl.append(criterion(pred, d["y"]).item())
                vcounts += np.array(ref_utils.calculate_tfpn(pred, d["y"]), dtype=np.int64)
        tfpn_valid.append(vcounts)
        loss_valid.append(float(np.mean(vl)))
        if len(loss_valid) > 1 and loss_valid[-1] < min(loss_valid[:-1]):                 # :525-527
            best_epoch = epoch
        scheduler.step(loss_valid[-1])                                                    # :529
    final = {k: v.detach().double().numpy() for k, v in model.state_dict().items()}
    return dict(ratio=ratio, init=init, step_losses=np.array(step_losses), step_graph=np.array(step_graph),
                tfpn_train=np.stack(tfpn_train), tfpn_valid=np.stack(tfpn_valid), loss_train=np.array(loss_train),
                loss_valid=np.array(loss_valid), lr=np.array(lrs), final_lr=optimizer.param_groups[0]["lr"],
                best_epoch=best_epoch, final=final)


if __name__ == "__main__":
    r32, r64 = run(torch.float32), run(torch.float64)
    out = {"hp_keys": np.array(sorted(HP)), "hp_vals": np.array([float(HP[k]) for k in sorted(HP)])}
    for i, s in enumerate(TRAIN):
        out[f"train{i}"] = np.array([s["reads"], s["seed"], int(s["permute"])])
    for i, s in enumerate(VALID):
        out[f"valid{i}"] = np.array([s["reads"], s["seed"], int(s["permute"])])
    for tag, r in (("32", r32), ("64", r64)):
        for k in ("step_losses", "step_graph", "tfpn_train", "tfpn_valid", "loss_train", "loss_valid", "lr"):
            out[f"{k}{tag}"] = r[k]
        out[f"final_lr{tag}"] = np.float64(r["final_lr"])
        out[f"best_epoch{tag}"] = np.int64(r["best_epoch"])
        out[f"ratio{tag}"] = np.float64(r["ratio"])
    # initial weights (the seeded nn.Linear init of THIS torch build; checked against the build's own seeded init)
    # and a strided sample of the final fp64 weights
    for k, v in r32["init"].items():
        out["init/" + k] = v.reshape(-1)[::13].copy()
    for k, v in r64["final"].items():
        out["final/" + k] = v.reshape(-1)[::13].copy()
    path = os.path.join(HERE, "train_loop_h128l2.npz")
    np.savez_compressed(path, **out)
    print("step losses fp64:", np.round(r64["step_losses"], 6))
    print("step losses fp32:", np.round(r32["step_losses"], 6))
    print("max rel diff 32 vs 64:", np.max(np.abs(r32["step_losses"] - r64["step_losses"]) / r64["step_losses"]))
    print("valid fp64:", r64["loss_valid"], "lr:", r64["lr"], "final lr", r64["final_lr"], "best", r64["best_epoch"])
    print("tfpn_train fp64:", r64["tfpn_train"].tolist(), "fp32:", r32["tfpn_train"].tolist())
    print("order:", r64["step_graph"].tolist(), "size KiB", os.path.getsize(path) / 1024)
