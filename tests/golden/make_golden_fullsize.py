"""Golden logits AND parameter gradients at the METRIC'S size (BASELINE config 2: R = 750 k reads, N = 1.5 M, E = 7,540,278, H = 128,
L = 8) from the CPU oracle in fp64, on the seeded synthetic graph / inputs / parameters that tests/test_gpu_parity.py rebuilds.

    python tests/golden/make_golden_fullsize.py              logits: oracle.model_forward under no_grad (~5 min on a 128-thread host,
                                                             ~25 [E,H] fp64 tensors alive at its peak: ~100 GB)
    python tests/golden/make_golden_fullsize.py --grads      gradients: oracle.bce_loss(oracle.model_forward(...)).backward() -- the autograd
                                                             form, i.e. what the reference's loss.backward() (train.py:253-257) computes --
                                                             with every GatedGCN layer under torch.utils.checkpoint so that only one layer's
                                                             autograd graph (~20 [E,H] fp64 tensors) is alive at a time (~300 GB instead of
                                                             ~1.3 TB; checkpointing re-runs the same deterministic CPU code: bit-identical
                                                             gradients, checked at R = 2 k by --selftest)

Stored (logits): the logits of every 97th edge (fp64) and the norm of all of them -- what test_full_size_logits_match_the_oracle compares
the HIP forward with on every run (GNM_FULL_ORACLE=1 makes the test run this oracle live and compare ALL logits instead).
Stored (gradients): the loss (fp64) and EVERY parameter gradient (826,033 values) as the fp32 rounding of the fp64 result (6e-8 relative,
far below the 2e-4 bar they are compared at) plus each tensor's fp64 norm -- test_full_size_gradients_match_the_oracle.
Everything else about the case is a seed.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gnnome_assembly_amd import synth          # noqa: E402
from oracle import gatedgcn_oracle as orc       # noqa: E402

R, H, L, SEED, STRIDE = 750000, 128, 8, 0, 97


# round 6: the reference's DEFAULT model shape (hyperparameters.py:8,13: dim_latent 256, num_gnn_layers 16) at the true chr19 size
# (R = 110 k reads), logits only:   python tests/golden/make_golden_fullsize.py --case h256l16_r110k
CASES = {"r750k": (750000, 128, 8, "fullsize_logits_r750k.npz"),
         "h256l16_r110k": (110000, 256, 16, "fullsize_logits_h256l16_r110k.npz")}


def _case(reads, H_=H, L_=L):
    src, dst, n = synth.make_graph(reads, SEED)
    inp = synth.make_inputs(src, dst, n, SEED)
    sd = {k: torch.from_numpy(np.asarray(v)).double() for k, v in synth.synth_state_dict(H_, L_, SEED).items()}
    return src, dst, n, inp, sd


def logits(case="r750k"):
    R, H, L, fname = CASES[case]
    src, dst, n, inp, sd = _case(R, H, L)
    t0 = time.perf_counter()
    with torch.no_grad():
        s = orc.model_forward(sd, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).double(),
                              torch.from_numpy(inp["pe"]).double()).squeeze(-1).numpy()
    secs = time.perf_counter() - t0
    idx = np.arange(0, s.size, STRIDE, dtype=np.int64)
    out = os.path.join(HERE, fname)
    np.savez_compressed(out, reads=R, H=H, L=L, seed=SEED, stride=STRIDE, edges=s.size, logits=s[idx],
                        norm2=float(np.linalg.norm(s)), mean=float(s.mean()), oracle_seconds=secs, threads=torch.get_num_threads())
    print(f"wrote {out}: {idx.size} of {s.size} logits (fp64 oracle, {secs:.0f} s on {torch.get_num_threads()} threads), |s|_2 = {np.linalg.norm(s):.6f}")


def oracle_grads(reads, checkpointed=True):
    """(loss, {key: fp64 gradient}) of the oracle's autograd form; checkpointed = one layer's graph alive at a time."""
    from torch.utils.checkpoint import checkpoint
    src, dst, n, inp, sd = _case(reads)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    s64, d64 = torch.from_numpy(src).long(), torch.from_numpy(dst).long()
    e_raw, pe = torch.from_numpy(inp["e"]).double(), torch.from_numpy(inp["pe"]).double()
    y, pw = torch.from_numpy(inp["y"]).double(), float(inp["pos_weight"])
    if not checkpointed:
        loss = orc.bce_loss(orc.model_forward(p, s64, d64, n, e_raw, pe), y, pw)
    else:       # orc.model_forward (full_graph.py:22-29), layer by layer under checkpoint
        h = pe @ p["linear_pe.weight"].t() + p["linear_pe.bias"]
        e = torch.relu(e_raw @ p["linear1_edge.weight"].t() + p["linear1_edge.bias"])
        e = e @ p["linear2_edge.weight"].t() + p["linear2_edge.bias"]
        for i in range(orc.num_layers_of(p)):
            h, e = checkpoint(lambda h_, e_, i_=i: orc.layer_forward(p, i_, s64, d64, n, h_, e_), h, e, use_reentrant=False)
            print(f"  forward layer {i} done", flush=True)
        scores = checkpoint(lambda h_, e_: orc.predictor_forward(p, s64, d64, h_, e_), h, e, use_reentrant=False)
        loss = orc.bce_loss(scores, y, pw)
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in p.items()}, int(src.size)


def grads():
    with open("/proc/meminfo") as f:
        avail = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable")) / 2**20
    if avail < 400:
        raise SystemExit(f"make_golden_fullsize --grads needs ~300 GB of host memory; {avail:.0f} GiB available here")
    t0 = time.perf_counter()
    loss, g, E = oracle_grads(R)
    secs = time.perf_counter() - t0
    out = os.path.join(HERE, "fullsize_grads_r750k.npz")
    arrays = {"grad::" + k: v.astype(np.float32) for k, v in g.items()}
    norms = {"norm::" + k: np.float64(np.linalg.norm(v)) for k, v in g.items()}
    np.savez_compressed(out, reads=R, H=H, L=L, seed=SEED, edges=E, loss=np.float64(loss), oracle_seconds=secs,
                        threads=torch.get_num_threads(), **arrays, **norms)
    print(f"wrote {out}: loss {loss:.9f}, {sum(v.size for v in g.values())} gradient values in {len(g)} tensors "
          f"(fp64 autograd oracle, layers checkpointed, {secs:.0f} s on {torch.get_num_threads()} threads)")


def selftest():
    """checkpointed == plain autograd, bit for bit, at a size the container runs in seconds"""
    l0, g0, _ = oracle_grads(2000, checkpointed=False)
    l1, g1, _ = oracle_grads(2000, checkpointed=True)
    assert l0 == l1 and all(np.array_equal(g0[k], g1[k]) for k in g0), "checkpointed gradients differ from plain autograd"
    print(f"selftest ok: loss {l0:.9f}, {len(g0)} gradient tensors bit-identical with and without checkpointing")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grads", action="store_true")
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--case", default="r750k", choices=sorted(CASES), help="which logits fixture to write")
    a = ap.parse_args()
    selftest() if a.selftest else grads() if a.grads else logits(a.case)
