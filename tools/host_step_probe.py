#!/usr/bin/env python3
"""Host-side duration of each call of one training step (no synchronisation anywhere): a call that takes as long as the GPU work
queued before it is a call that blocks the host.  gpurun_out/host_step_probe.txt"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp, synth
    dev = torch.device("cuda:0")
    R, H, L = int(os.environ.get("READS", "750000")), 128, 8
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
    opt = dp.make_adam(model.parameters(), 1e-3)
    names = ("zero", "forward", "loss", "backward", "all_reduce", "optimizer")
    lines = []
    for it in range(6):
        if it == 3:
            torch.cuda.synchronize()
        t = [time.perf_counter()]
        flat.zero_(); t.append(time.perf_counter())
        s = model(g, None, e, pe); t.append(time.perf_counter())
        loss = crit(s.squeeze(-1), y); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        flat.all_reduce_mean(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        lines.append(f"step {it}: host ms " + ", ".join(f"{k} {(b - a) * 1e3:.2f}" for k, a, b in zip(names, t, t[1:])) + f"; total {(t[-1] - t[0]) * 1e3:.2f}")
    # device-side: events at the first and last launch of every step; the time between one step's last event and the next
    # step's first is GPU time no kernel of ours accounts for
    torch.cuda.synchronize()
    ev = []
    for it in range(8):
        a, b, c, d = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        a.record()
        flat.zero_()
        s = model(g, None, e, pe)
        b.record()
        loss = crit(s.squeeze(-1), y)
        loss.backward()
        c.record()
        flat.all_reduce_mean()
        opt.step()
        d.record()
        ev.append((a, b, c, d))
    torch.cuda.synchronize()
    for k in range(1, 8):
        a, b, c, d = ev[k]
        lines.append(f"device step {k}: forward {a.elapsed_time(b):.2f} ms, loss+backward {b.elapsed_time(c):.2f}, optimizer {c.elapsed_time(d):.2f}, "
                     f"from the previous step's last event to this step's first {ev[k - 1][3].elapsed_time(a):.3f}")
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    lines.append(f"final synchronize: {(time.perf_counter() - t0) * 1e3:.1f} ms (the GPU work the host was ahead by)")
    out = "\n".join(lines)
    print(out)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    open(os.path.join(REPO, "gpurun_out", "host_step_probe.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
