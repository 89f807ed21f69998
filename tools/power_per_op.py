#!/usr/bin/env python3
"""Socket power, shader clock and ENERGY per launch of every C-ABI op of the training step: one step is recorded (the last call of
each op with its arguments), then each recorded call is replayed alone in a loop for ~2.5 s under the hwmon sampler of
tools/power_probe.py.  A kernel that sits at the 1400 W package cap with the clock below 2.4 GHz is bound by the energy it
spends; joules per launch x launches per step says where the step's ~220 J go.  (The replays run on whatever the buffers hold
after the step; in-place ops keep accumulating, which does not change what the kernels do per launch.)
gpurun_out/power_per_op.txt        READS / HIDDEN / LAYERS / GNM_MATMUL from the environment"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from power_probe import Sampler  # noqa: E402


def main():
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp, engine, synth
    dev = torch.device("cuda:0")
    R, H, L = int(os.environ.get("READS", "750000")), int(os.environ.get("HIDDEN", "128")), int(os.environ.get("LAYERS", "8"))
    src, dst, n = synth.make_graph(R, seed=0)
    inp = synth.make_inputs(src, dst, n, seed=0)
    g = G.AssemblyGraph(src, dst, n).to(dev)
    g.index()
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
    model.to(dev)
    model.flatten_parameters()
    e, pe, y = (torch.from_numpy(inp[k]).to(dev) for k in ("e", "pe", "y"))
    crit = G.BCEWithLogitsLoss(float(inp["pos_weight"]))
    flat = dp.FlatGradients(model.parameters(), direct_write=True)

    def step():
        flat.zero_()
        loss = crit(model(g, None, e, pe).squeeze(-1), y)
        loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rec, count = {}, {}
    orig = engine._call

    def recording(name, *args, tag=None):
        key = tag or name
        rec[key] = (getattr(G._lib.load(), name), args)
        count[key] = count.get(key, 0) + 1
        return orig(name, *args, tag=tag)

    engine._call = recording
    keep = model(g, None, e, pe)                    # the forward's saved tensors stay alive through `keep`
    loss = crit(keep.squeeze(-1), y)
    flat.zero_()
    loss.backward(retain_graph=True)
    torch.cuda.synchronize()
    engine._call = orig

    smp = Sampler()
    smp.start()
    rows = []
    for key, (fn, args) in rec.items():
        for _ in range(3):
            fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if ms * count[key] < 0.15:                 # below 0.1 % of the step
            continue
        nb = max(10, int(150.0 / max(ms, 1e-3)))   # ~150 ms of launches between synchronisations
        smp.samples = []
        t0 = time.perf_counter()
        k = 0
        torch.cuda.synchronize()
        smp.on = True
        while time.perf_counter() - t0 < 2.5:
            for _ in range(nb):
                fn(*args)
            torch.cuda.synchronize()
            k += nb
        smp.on = False
        dt = (time.perf_counter() - t0) / k * 1e3
        s = smp.samples[len(smp.samples) // 4:]    # the first quarter: the power average still ramps
        if not s:
            continue
        w = sum(x[0] for x in s) / len(s)
        f = sum(x[1] for x in s) / len(s)
        rows.append((key, count[key], dt, w, f, dt * 1e-3 * w))
    smp.done = True
    rows.sort(key=lambda r: -r[1] * r[5])
    tot_j = sum(r[1] * r[5] for r in rows)
    tot_ms = sum(r[1] * r[2] for r in rows)
    out = [f"# R={R} H={H} L={L} matmul={G._lib.get_matmul_mode()}; hwmon {smp.pw}; each op replayed alone for 2.5 s",
           f"{'op':34s} {'per step':>8s} {'ms/launch':>10s} {'power W':>8s} {'sclk MHz':>9s} {'J/launch':>9s} {'J/step':>8s} {'share':>6s}"]
    for key, c, ms, w, f, j in rows:
        out.append(f"{key:34s} {c:8d} {ms:10.3f} {w:8.0f} {f:9.0f} {j:9.3f} {c * j:8.2f} {c * j / tot_j:6.1%}")
    out.append(f"{'sum':34s} {'':8s} {tot_ms:10.2f} {'':8s} {'':9s} {'':9s} {tot_j:8.1f}")
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "power_per_op.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
