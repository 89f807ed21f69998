"""Training-harness counterpart of the reference's full-graph branch (train.py:115-533 with
batch_size_train <= 1): per-graph Adam steps, validation under no_grad/eval, ReduceLROnPlateau,
best-model + checkpoint files with the reference's key schema.  SURVEY.md section 8f row 2.

Differences from the reference, all deliberate:
  * graphs, features and the index stay resident on the device (the reference re-uploads every
    step, train.py:245-251);
  * TP/TN/FP/FN are accumulated on the device and read back once per epoch (the reference's
    utils.calculate_tfpn forces four .item() syncs per graph, utils.py:217-223);
  * under torch.distributed (one process per GPU) the W graphs of a step contribute the MEAN of
    their gradients through one RCCL all-reduce (dp.FlatGradients); single-process runs reproduce the
    reference's one-step-per-graph sequence;
  * the ClusterGCN mini-batch branch (train.py:282-343,428-486; batch_size_* > 1) runs on this
    package's own partitioner (cluster.py: METIS is part of DGL and not available), otherwise with
    the reference's loop: a fresh partition per graph and epoch with a random number of clusters in
    [num_parts-100, num_parts+100), shuffled batches of clusters, one Adam step per batch;
  * wandb and the data pipeline are out of scope.
`calculate_metrics` keeps the reference's naming, in which "precision" and "recall" are swapped
(utils.py:227-234)."""
from __future__ import annotations

import copy
import os
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import cluster, dp, models

__all__ = ["get_hyperparameters", "GraphSample", "tfpn_counts", "calculate_metrics", "train", "save_checkpoint"]


def get_hyperparameters() -> Dict:
    """hyperparameters.py:3-34 with the full-graph branch selected (batch_size_* = 1)."""
    return {
        "seed": 0, "lr": 1e-3, "num_epochs": 100, "dim_latent": 256, "node_features": 1, "edge_features": 2,
        "hidden_edge_features": 16, "hidden_edge_scores": 64, "num_gnn_layers": 16, "nb_pos_enc": 16,
        "batch_size_train": 1, "batch_size_eval": 1, "patience": 2, "decay": 0.95, "batch_norm": True,
        # only read when batch_size_* > 1 (hyperparameters.py:15-18 has 500 / 500 / 50 / 50)
        "num_parts_metis_train": 500, "num_parts_metis_eval": 500, "partition_method": "locality",
    }


@dataclass
class GraphSample:
    """One training graph resident on the device: what train.py:245-254 pulls out of a DGLGraph."""
    graph: object                 # AssemblyGraph on the device
    e: torch.Tensor               # [E,2] z-scored edge features (edge-id order)
    pe: torch.Tensor              # [N, nb_pos_enc+2] = in_deg | out_deg | pe
    y: torch.Tensor               # [E] float labels
    x: Optional[torch.Tensor] = None


def tfpn_counts(edge_predictions: torch.Tensor, edge_labels: torch.Tensor) -> torch.Tensor:
    """[TP, TN, FP, FN] as a device int64 tensor (utils.calculate_tfpn without the .item() syncs)."""
    p = torch.round(torch.sigmoid(edge_predictions))
    return torch.stack([((p == 1) & (edge_labels == 1)).sum(), ((p == 0) & (edge_labels == 0)).sum(),
                        ((p == 1) & (edge_labels == 0)).sum(), ((p == 0) & (edge_labels == 1)).sum()])


def calculate_metrics(TP, TN, FP, FN):
    """utils.calculate_metrics (utils.py:226-240), including its swapped precision/recall names."""
    recall = TP / (TP + FP) if (TP + FP) else 0
    precision = TP / (TP + FN) if (TP + FN) else 0
    f1 = TP / (TP + 0.5 * (FP + FN)) if (TP + 0.5 * (FP + FN)) else 0
    accuracy = (TP + TN) / (TP + TN + FP + FN)
    return accuracy, precision, recall, f1


def save_checkpoint(epoch, model, optimizer, loss_train, loss_valid, out, directory="checkpoints"):
    """train.py:28-58: same dict keys, same file name."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{out}.pt")
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optim_state_dict": optimizer.state_dict(),
                "loss_train": loss_train, "loss_valid": loss_valid}, path)
    return path


def pos_to_neg_ratio(samples: Sequence[GraphSample]) -> float:
    """train.py:181: dataset mean of #(y==1)/#(y==0)."""
    r = [((s.y == 1).sum() / (s.y == 0).sum()) for s in samples]
    return float(torch.stack(r).mean().item())


def _cluster_batches(s: GraphSample, num_parts: int, batch_size: int, method: str):
    """The sub-graphs of one pass over a graph in mini-batch mode; features ride on ndata / edata exactly
    as the reference keeps them on the DGLGraph (train.py:298-307)."""
    g = s.graph
    g.ndata = dict(g.ndata, pe=s.pe)
    g.edata = dict(g.edata, e=s.e, y=s.y)
    part = cluster.partition_graph(g, num_parts, method)
    return cluster.ClusterBatchLoader(g, part, batch_size, shuffle=True)


@dataclass
class History:
    loss_train: List[float] = field(default_factory=list)
    loss_valid: List[float] = field(default_factory=list)
    lr: List[float] = field(default_factory=list)
    metrics_train: List[tuple] = field(default_factory=list)
    metrics_valid: List[tuple] = field(default_factory=list)
    step_losses: List[float] = field(default_factory=list)      # every optimizer step's loss, in order (this rank's)
    step_graph: List[int] = field(default_factory=list)         # ... and the index of the graph it was taken on
    tfpn_train: List[tuple] = field(default_factory=list)       # per epoch: (TP, TN, FP, FN) summed over the graphs
    tfpn_valid: List[tuple] = field(default_factory=list)
    best_epoch: int = -1
    final_lr: float = 0.0


def train(train_samples: Sequence[GraphSample], valid_samples: Sequence[GraphSample], out: str = "model",
          hyperparameters: Optional[Dict] = None, workdir: str = ".", verbose: bool = True,
          hooks: Optional[Dict] = None):
    """Full-graph training loop (train.py:232-281,379-529).  Returns (model, best_state_dict, History).
    Under torch.distributed every rank passes ITS shard of the training AND of the validation graphs
    (dp.shard_graphs; a shard may be empty for validation): the W graphs of a step contribute the mean of their
    gradients, epoch / validation losses and TP..FN counts are summed over the ranks, rank 0 writes the files.
    `hooks` (observers for tests and logging, never needed for training): "after_exchange"(epoch, it, flat) right
    after the gradient exchange of a full-graph step, "after_epoch"(epoch, model) at the end of every epoch;
    "model_factory"(hp) -> nn.Module and "criterion_factory"(pos_weight) -> loss module replace the HIP model / loss (the
    CPU tier drives THIS loop under eight gloo ranks with a CPU stand-in model: tests/test_dp_gloo.py)."""
    hooks = hooks or {}
    hp = dict(get_hyperparameters())
    hp.update(hyperparameters or {})
    seed = hp["seed"]
    random.seed(seed)
    torch.manual_seed(seed)                                                     # utils.set_seed
    dev = train_samples[0].pe.device
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ratio = pos_to_neg_ratio(train_samples)                                     # train.py:181
    if world > 1:
        t = torch.tensor([ratio * len(train_samples), float(len(train_samples))], device=dev, dtype=torch.float64)
        dist.all_reduce(t)
        ratio = float(t[0] / t[1])
    if "model_factory" in hooks:
        model = hooks["model_factory"](hp).to(dev)
    else:
        model = models.GraphGatedGCNModel(hp["node_features"], hp["edge_features"], hp["dim_latent"],
                                          hp["hidden_edge_features"], hp["num_gnn_layers"], hp["hidden_edge_scores"],
                                          hp["batch_norm"], hp["nb_pos_enc"]).to(dev)            # train.py:195-198
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    best_state = copy.deepcopy(model.state_dict())                                              # train.py:203
    models.flatten_parameters(model)
    flat = dp.FlatGradients(model.parameters(), direct_write=True)       # the loop below zero_()s before every backward
    optimizer = dp.make_adam(model.parameters(), hp["lr"])                               # train.py:209
    criterion = (hooks["criterion_factory"](1.0 / ratio) if "criterion_factory" in hooks
                 else models.BCEWithLogitsLoss(pos_weight=1.0 / ratio))                         # train.py:210-211
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=hp["decay"],
                                                           patience=hp["patience"])             # train.py:212
    hist = History()
    order = list(range(len(train_samples)))
    model_path = os.path.join(workdir, "pretrained", f"model_{out}.pt")
    for epoch in range(hp["num_epochs"]):
        random.shuffle(order)                                                                   # train.py:238
        model.train()
        loss_sum = torch.zeros((), device=dev, dtype=torch.float64)
        counts = torch.zeros(4, device=dev, dtype=torch.int64)
        ep_losses = []                       # device scalars; read back once per epoch
        # every rank takes the same number of optimizer steps: shards may be uneven (dp.shard_graphs), the
        # shorter ranks pad with zero-contribution steps so that the gradient collectives stay matched
        nsteps = dp.steps_per_epoch(len(order), dev)
        for it in range(nsteps):
            s = train_samples[order[it]] if it < len(order) else None
            if hp["batch_size_train"] <= 1:                                                     # full graph
                flat.zero_()
                if s is not None:
                    pred = model(s.graph, s.x, s.e, s.pe).squeeze(-1)                           # train.py:252-253
                    loss = criterion(pred, s.y)
                    loss.backward()
                flat.all_reduce_mean(contributed=s is not None)
                if "after_exchange" in hooks:
                    hooks["after_exchange"](epoch, it, flat)
                optimizer.step()                                                                # train.py:256-258
                if s is not None:
                    loss_sum += loss.detach().double()
                    ep_losses.append(loss.detach())
                    hist.step_graph.append(order[it])
                    counts += tfpn_counts(pred.detach(), s.y)
            else:                                                                               # train.py:282-343
                lo = max(1, hp["num_parts_metis_train"] - 100)
                nparts = int(torch.randint(lo, hp["num_parts_metis_train"] + 100, (1,)).item())  # train.py:291
                gl = torch.zeros((), device=dev, dtype=torch.float64)
                nb = 0
                loader = _cluster_batches(s, nparts, hp["batch_size_train"], hp["partition_method"]) if s is not None else []
                nbatches = dp.steps_per_epoch(len(loader), dev)          # the ranks' cluster counts differ
                batches = iter(loader)
                for _ in range(nbatches):
                    sub = next(batches, None)
                    flat.zero_()
                    if sub is not None:
                        pred = model(sub, None, sub.edata["e"], sub.ndata["pe"]).squeeze(-1)    # train.py:306
                        loss = criterion(pred, sub.edata["y"])
                        loss.backward()
                    flat.all_reduce_mean(contributed=sub is not None)
                    optimizer.step()
                    if sub is not None:
                        gl += loss.detach().double()
                        nb += 1
                        counts += tfpn_counts(pred.detach(), sub.edata["y"])
                if s is not None:
                    loss_sum += gl / max(nb, 1)                                                 # train.py:330
        n_train = torch.tensor(float(len(order)), device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(loss_sum); dist.all_reduce(n_train); dist.all_reduce(counts)        # noqa: E702
        train_loss = float(loss_sum / n_train)
        hist.loss_train.append(train_loss)
        if ep_losses:
            hist.step_losses += torch.stack(ep_losses).double().cpu().tolist()
        hist.tfpn_train.append(tuple(int(c) for c in counts.tolist()))
        hist.metrics_train.append(calculate_metrics(*hist.tfpn_train[-1]))
        # ---- validation (train.py:385-511): eval mode, no_grad, no activations kept ----
        model.eval()
        vloss = torch.zeros((), device=dev, dtype=torch.float64)
        vcounts = torch.zeros(4, device=dev, dtype=torch.int64)
        with torch.no_grad():
            for s in valid_samples:
                if hp["batch_size_eval"] <= 1:
                    pred = model(s.graph, s.x, s.e, s.pe).squeeze(-1)
                    vloss += criterion(pred, s.y).double()
                    vcounts += tfpn_counts(pred, s.y)
                else:                                                                           # train.py:428-486
                    gl = torch.zeros((), device=dev, dtype=torch.float64)
                    nb = 0
                    for sub in _cluster_batches(s, hp["num_parts_metis_eval"], hp["batch_size_eval"],
                                                hp["partition_method"]):
                        pred = model(sub, None, sub.edata["e"], sub.ndata["pe"]).squeeze(-1)
                        gl += criterion(pred, sub.edata["y"]).double()
                        nb += 1
                        vcounts += tfpn_counts(pred, sub.edata["y"])
                    vloss += gl / max(nb, 1)
        n_val = torch.tensor(float(len(valid_samples)), device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(vloss); dist.all_reduce(n_val); dist.all_reduce(vcounts)            # noqa: E702
        val_loss = float(vloss / n_val.clamp(min=1))
        hist.loss_valid.append(val_loss)
        hist.tfpn_valid.append(tuple(int(c) for c in vcounts.tolist()))
        hist.metrics_valid.append(calculate_metrics(*hist.tfpn_valid[-1]) if int(vcounts.sum()) else None)
        hist.lr.append(optimizer.param_groups[0]["lr"])
        if len(hist.loss_valid) > 1 and hist.loss_valid[-1] < min(hist.loss_valid[:-1]):        # train.py:525-527
            best_state = copy.deepcopy(model.state_dict())
            hist.best_epoch = epoch
            if rank == 0:
                os.makedirs(os.path.dirname(model_path), exist_ok=True)
                torch.save(best_state, model_path)
        if rank == 0:
            save_checkpoint(epoch, model, optimizer, train_loss, val_loss, out, os.path.join(workdir, "checkpoints"))
        scheduler.step(val_loss)                                                                # train.py:529
        hist.final_lr = optimizer.param_groups[0]["lr"]
        if "after_epoch" in hooks:
            hooks["after_epoch"](epoch, model)
        if verbose and rank == 0:
            print(f"epoch {epoch}: train loss {train_loss:.4f}  valid loss {val_loss:.4f}  lr {hist.lr[-1]:.2e}")
    return model, best_state, hist
