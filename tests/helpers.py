"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np
import torch

from gnnome_assembly_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# The parity bar (BASELINE.json north_star: "within 1e-4 rel fp32"; SURVEY.md section 7
# "parity metric definition"): allclose(rtol=1e-4, atol=1e-5) AND norm-relative <= 1e-4.
RTOL = 1e-4
ATOL = 1e-5


def load_case(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    H, L, seed = int(z["H"]), int(z["L"]), int(z["seed"])
    sd = synth.synth_state_dict(H, L, seed)
    return z, sd, H, L, bool(z["batch_norm"])


def sd_to_torch(sd, dtype=torch.float32, requires_grad=False):
    return {k: torch.from_numpy(np.asarray(v)).to(dtype).clone().requires_grad_(requires_grad)
            for k, v in sd.items()}


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_parity(got, want, what, rtol=RTOL, atol=ATOL, l2=1e-4):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.all(np.isfinite(got)), f"{what}: non-finite values"
    r = rel_l2(got, want)
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert r <= l2 and not bad.any(), (
        f"{what}: rel_l2={r:.3e} max_abs={np.abs(got - want).max():.3e} "
        f"violations={int(bad.sum())}/{bad.size}")
