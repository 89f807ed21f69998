"""Golden logits at the METRIC'S size (BASELINE config 2: R = 750 k reads, N = 1.5 M, E = 7,540,278, H = 128, L = 8): the CPU
oracle's fp64 forward (oracle.model_forward under no_grad, ~5 min on a 128-thread host, ~25 [E,H] fp64 tensors alive at its peak) on
the seeded synthetic graph / inputs / parameters that tests/test_gpu_parity.py::test_full_size_logits_match_the_oracle rebuilds.  Stored:
the logits of every 97th edge (fp64) -- what the test compares the HIP forward with on every run; GNM_FULL_ORACLE=1 makes the test
run this oracle live and compare ALL logits instead.  Everything else about the case is a seed.

    python tests/golden/make_golden_fullsize.py          (needs ~100 GB of host memory; run it where there is that much)
"""
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


def main():
    src, dst, n = synth.make_graph(R, SEED)
    inp = synth.make_inputs(src, dst, n, SEED)
    sd = {k: torch.from_numpy(np.asarray(v)).double() for k, v in synth.synth_state_dict(H, L, SEED).items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        s = orc.model_forward(sd, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).double(),
                              torch.from_numpy(inp["pe"]).double()).squeeze(-1).numpy()
    secs = time.perf_counter() - t0
    idx = np.arange(0, s.size, STRIDE, dtype=np.int64)
    out = os.path.join(HERE, "fullsize_logits_r750k.npz")
    np.savez_compressed(out, reads=R, H=H, L=L, seed=SEED, stride=STRIDE, edges=s.size, logits=s[idx],
                        norm2=float(np.linalg.norm(s)), mean=float(s.mean()), oracle_seconds=secs, threads=torch.get_num_threads())
    print(f"wrote {out}: {idx.size} of {s.size} logits (fp64 oracle, {secs:.0f} s on {torch.get_num_threads()} threads), |s|_2 = {np.linalg.norm(s):.6f}")


if __name__ == "__main__":
    main()
