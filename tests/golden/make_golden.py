#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own code.

Run in the build container only (it reads /root/reference, which does not exist on the
GPU box):      python tests/golden/make_golden.py

How: tests/golden/dgl_standin (a test-only stand-in for the 5 DGL builtins the hot path
uses; DGL itself cannot be installed here) goes first on sys.path, /root/reference
second; the reference's `models` / `layers` / `utils` packages are then imported
UNMODIFIED and run on torch-CPU.  Everything torch does (Linear, BatchNorm1d, LayerNorm,
relu, sigmoid, BCEWithLogitsLoss, autograd, Adam) is the real reference + real torch.
Only data (inputs and expected outputs) is written; no reference source is copied.

Fixtures (one .npz per case; parameters are regenerated from `seed` by
gnnome_assembly_amd.synth.synth_state_dict, so they are not stored):
  case = {graph: tiny|small} x {cfg: h64l1 | h128l8 | h32l2ln} x {seed 0,1}
         + the reference's default width / depth: small_h256l2_s0, tiny_h256l2_s1, tiny_h256l16_s0
  stored: src dst n e_raw pe y pos_weight | scores32 scores64 loss32 loss64 |
          per-layer h,e (L<=2, fp64) | grads64 (full for small models, strided sample
          for h128l8) | adam-step params (same) | 3-step loss sequence | tfpn | eval==train
  pe_pagerank.npz: utils.add_positional_encoding output for the small graph ("next" row).
"""
import os
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
from gnnome_assembly_amd import synth  # noqa: E402

torch.set_num_threads(4)
GRAD_STRIDE = 29   # sampling stride for the big model's gradients
ROW_STRIDE = 37    # row sampling stride for per-layer outputs of the small graph

CFGS = {
    "h64l1": dict(H=64, L=1, bn=True),
    "h128l8": dict(H=128, L=8, bn=True),
    "h32l2ln": dict(H=32, L=2, bn=False),
    # round 6: the reference's own default width (hyperparameters.py:8 dim_latent 256) and depth (:13 num_gnn_layers 16)
    "h256l2": dict(H=256, L=2, bn=True, grad_stride=211, tiny_row_stride=4),
    "h256l16": dict(H=256, L=16, bn=True, grad_stride=211),
}
SAMPLED = ("h128l8", "h256l2", "h256l16")      # gradients / Adam-step parameters stored every GRAD_STRIDE-th element (size)


def build_graph(kind, seed):
    if kind == "tiny":
        src, dst, n = synth.tiny_edge_case_graph(seed)
    else:
        src, dst, n = synth.make_graph(1000, seed, permute_edge_ids=(seed % 2 == 1))
    return src, dst, n


def ref_model(cfg, sd_np, dtype):
    m = models.GraphGatedGCNModel(1, 2, cfg["H"], 16, cfg["L"], 64, cfg["bn"], 16)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(dtype)


def sample(name, arr, full, stride=GRAD_STRIDE):
    a = arr.detach().double().numpy().reshape(-1)
    return (a if full else a[::stride]).copy()   # copy: later optimizer steps mutate the parameter


def run_case(kind, cfg_name, seed):
    cfg = CFGS[cfg_name]
    src, dst, n = build_graph(kind, seed)
    inp = synth.make_inputs(src, dst, n, seed)
    sd_np = synth.synth_state_dict(cfg["H"], cfg["L"], seed)
    out = dict(src=src, dst=dst, n=np.int64(n), e_raw=inp["e"], pe=inp["pe"], y=inp["y"],
               pos_weight=inp["pos_weight"], seed=np.int64(seed), H=np.int64(cfg["H"]),
               L=np.int64(cfg["L"]), batch_norm=np.bool_(cfg["bn"]), grad_stride=np.int64(cfg.get("grad_stride", GRAD_STRIDE)))
    gstride = cfg.get("grad_stride", GRAD_STRIDE)
    full = cfg_name not in SAMPLED
    if cfg["H"] != 128:
        out["grad_sampled"] = np.bool_(not full)      # (H = 128 fixtures predate the flag: tests/helpers.py grad_stride_of)
    for dtype, tag in ((torch.float32, "32"), (torch.float64, "64")):
        g = dgl.graph((src, dst), num_nodes=n)
        m = ref_model(cfg, sd_np, dtype)
        m.train()
        x = torch.from_numpy(inp["x"]).to(dtype)
        e = torch.from_numpy(inp["e"]).to(dtype)
        pe = torch.from_numpy(inp["pe"]).to(dtype)
        y = torch.from_numpy(inp["y"]).to(dtype)
        pw = torch.tensor([float(inp["pos_weight"])], dtype=dtype)
        crit = torch.nn.BCEWithLogitsLoss(pos_weight=pw)             # train.py:210-211
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)              # train.py:209
        losses = []
        for step in range(3):                                        # train.py:252-258
            pred = m(g, x, e, pe)
            loss = crit(pred.squeeze(-1), y)
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out["scores" + tag] = pred.detach().numpy().copy()
                out["loss" + tag] = np.float64(loss.item())
                if tag == "64":
                    for k, p in m.named_parameters():
                        out["grad/" + k] = sample(k, p.grad, full, gstride)
                    TP, TN, FP, FN = ref_utils.calculate_tfpn(pred.squeeze(-1), y)  # utils.py:217-223
                    out["tfpn"] = np.array([TP, TN, FP, FN], dtype=np.int64)
                    out["metrics"] = np.array(ref_utils.calculate_metrics(TP, TN, FP, FN))
            opt.step()
            if step == 0 and tag == "64":
                for k, p in m.named_parameters():
                    out["adam/" + k] = sample(k, p, full, gstride)
            losses.append(loss.item())
        out["loss_seq" + tag] = np.array(losses, dtype=np.float64)
        if tag == "64":
            # eval-mode output == train-mode output (no running stats: gated_gcn_full.py:55-56)
            m2 = ref_model(cfg, sd_np, dtype)
            m2.eval()
            with torch.no_grad():
                se = m2(g, x, e, pe)
            m2.train()
            with torch.no_grad():
                st = m2(g, x, e, pe)
            out["eval_equals_train"] = np.bool_(torch.equal(se, st))
            if cfg["L"] <= 2:
                # per-layer (h, e): processor.py:16-17, each GatedGCN_1d.forward output
                hh = m2.linear_pe(pe)
                ee = m2.linear2_edge(torch.relu(m2.linear1_edge(e)))
                with torch.no_grad():
                    for i, conv in enumerate(m2.gnn.convs):
                        hh, ee = conv(g, hh, ee)
                        # full for the tiny graph; every ROW_STRIDE-th row otherwise (size)
                        rs = cfg.get("tiny_row_stride", 1) if kind == "tiny" else ROW_STRIDE
                        out[f"layer{i}/h"] = hh.numpy()[::rs].copy()
                        out[f"layer{i}/e"] = ee.numpy()[::rs].copy()
                out["row_stride"] = np.int64(cfg.get("tiny_row_stride", 1) if kind == "tiny" else ROW_STRIDE)
    # dataset-level ratio of train.py:181 for this single graph
    yt = torch.from_numpy(inp["y"])
    out["pos_to_neg_ratio"] = np.float64(((yt == 1).sum() / (yt == 0).sum()).item())
    path = os.path.join(HERE, f"{kind}_{cfg_name}_s{seed}.npz")
    np.savez_compressed(path, **out)
    print(f"{os.path.basename(path)}: E={src.size} N={n} loss64={out['loss64']:.9f} "
          f"loss32={out['loss32']:.9f} |s32-s64|max={np.abs(out['scores32']-out['scores64']).max():.2e} "
          f"size={os.path.getsize(path)/1024:.0f} KiB")


def run_pe():
    src, dst, n = synth.make_graph(1000, 0)
    g = dgl.graph((src, dst), num_nodes=n)
    g = ref_utils.add_positional_encoding(g, 16)                     # utils.py:97-140
    np.savez_compressed(os.path.join(HERE, "pe_pagerank.npz"), src=src, dst=dst, n=np.int64(n),
                        pe=g.ndata["pe"].numpy(), in_deg=g.ndata["in_deg"].numpy(),
                        out_deg=g.ndata["out_deg"].numpy())
    print("pe_pagerank.npz written")


def cases():
    for kind in ("tiny", "small"):
        for cfg_name in CFGS:
            for seed in (0, 1):
                if cfg_name == "h128l8" and kind == "tiny" and seed == 1:
                    continue
                if cfg_name == "h256l2" and not (kind == "small" and seed == 0 or kind == "tiny" and seed == 1):
                    continue
                if cfg_name == "h256l16" and not (kind == "tiny" and seed == 0):
                    continue
                yield kind, cfg_name, seed


if __name__ == "__main__":
    only = sys.argv[1:]                 # e.g. `make_golden.py h256`: only the cases whose name contains an argument
    for kind, cfg_name, seed in cases():
        if only and not any(o in f"{kind}_{cfg_name}_s{seed}" for o in only):
            continue
        run_case(kind, cfg_name, seed)
    if not only:
        run_pe()
