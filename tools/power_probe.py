#!/usr/bin/env python3
"""Socket power and shader clock while the training step (or one half of it) runs in a loop.  The package power cap of the
MI355X is 1400 W; a kernel that sits at the cap runs at whatever shader clock the cap leaves it, so its duration is set by the
ENERGY it spends, not by the HBM or matrix-core peaks.  Samples come from the amdgpu hwmon files (power1_average in uW,
freq1_input in Hz), read every ~5 ms on a host thread; rocm-smi is the fallback.  gpurun_out/power_probe.txt

  python tools/power_probe.py [train|forward|backward] [seconds]        READS / HIDDEN / LAYERS from the environment
"""
import glob
import os
import subprocess
import sys
import threading
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def hwmon_files():
    """(power file, shader-clock file) of the GPU torch calls cuda:0 -- the box's sysfs lists every card of the node, so the card is
    matched by PCI address; without a match: the first card that has both files."""
    want = None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pass
    found = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = [os.path.join(d, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, n))]
        f = os.path.join(d, "freq1_input")
        if p and os.path.exists(f):
            found.append((os.path.realpath(os.path.dirname(os.path.dirname(d))), p[0], f))
    for dev, p, f in found:
        if want and dev.endswith(want):
            return p, f
    if want and found:
        print(f"power_probe: no card at PCI {want} among {[x[0][-12:] for x in found]}; using the first", file=sys.stderr)
    return (found[0][1], found[0][2]) if found else (None, None)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.pw, self.fq = hwmon_files()
        self.samples, self.on, self.done = [], False, False

    def read(self):
        if self.pw:
            return int(open(self.pw).read()) / 1e6, int(open(self.fq).read()) / 1e6
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        w = mhz = float("nan")
        for line in out.splitlines():
            if "Package Power" in line:
                w = float(line.rsplit(":", 1)[1])
            if "sclk" in line and "Mhz" in line:
                mhz = float(line.split("(")[1].split("Mhz")[0])
        return w, mhz

    def run(self):
        while not self.done:
            if self.on:
                self.samples.append(self.read())
            time.sleep(0.005 if self.pw else 0.2)


def summary(name, s, ms):
    if not s:
        return f"{name}: no samples"
    w = sorted(x[0] for x in s)
    f = sorted(x[1] for x in s)
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
    below = sum(1 for x in f if x < 2200) / len(f)
    return (f"{name}: {ms:8.2f} ms per pass | power W mean {sum(w) / len(w):6.0f}  p10 {q(w, .1):6.0f}  p50 {q(w, .5):6.0f}  p90 {q(w, .9):6.0f}"
            f" | sclk MHz mean {sum(f) / len(f):5.0f}  p10 {q(f, .1):5.0f}  p50 {q(f, .5):5.0f}  p90 {q(f, .9):5.0f} | samples {len(s)}, below 2200 MHz {below:.0%}")


def main():
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import dp, synth
    what = sys.argv[1] if len(sys.argv) > 1 else "train"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
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
    opt = dp.make_adam(model.parameters(), 1e-3)

    def train():
        flat.zero_()
        loss = crit(model(g, None, e, pe).squeeze(-1), y)
        loss.backward()
        flat.all_reduce_mean()
        opt.step()

    def forward():
        with torch.no_grad():
            model(g, None, e, pe)

    fn = {"train": train, "forward": forward}[what]
    smp = Sampler()
    smp.start()
    idle = smp.read()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    smp.on = True
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < secs:
        fn()
        torch.cuda.synchronize()
        k += 1
    ms = (time.perf_counter() - t0) / k * 1e3
    smp.on = False
    smp.done = True
    src_ = "hwmon " + smp.pw if smp.pw else "rocm-smi"
    out = [f"# {what} pass, R={R} H={H} L={L} matmul={G._lib.get_matmul_mode()}, {secs:.0f} s loop, samples from {src_}; idle before: {idle[0]:.0f} W, {idle[1]:.0f} MHz",
           summary(what, smp.samples, ms)]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "power_probe.txt"), "a") as fh:
        fh.write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
