#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/collect_traffic.sh into per-launch HBM bytes.

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide (16 B/lane) coalesced streaming read, so the read side is doubled.  WRITE_SIZE is
taken as is.  Writes profiles/<name>.json: {kernel: {fetch_gb, write_gb, total_gb, launches}}."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_assembly_amd._lib import csrc_sha   # noqa: E402  (pure python: hashes csrc/ + include/gnm.h)
d, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "n_f": 0, "n_w": 0})
for c, key, nk in (("FETCH_SIZE", "fetch_kib", "n_f"), ("WRITE_SIZE", "write_kib", "n_w")):
    for r in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void gnm::", "").replace("gnm::", "")
        if k.startswith("edge_bwd_chain_k"):
            k = "edge_bwd_chain_k"
        res[k][key] += float(r["Counter_Value"])
        res[k][nk] += 1
table = {}
for k, v in res.items():
    if not v["n_f"] or not v["n_w"]:
        continue
    f = 2.0 * v["fetch_kib"] * 1024 / v["n_f"]     # gfx950: FETCH_SIZE counts 64 B per 128-B request
    w = v["write_kib"] * 1024 / v["n_w"]
    table[k] = {"fetch_gb": f / 1e9, "write_gb": w / 1e9, "total_gb": (f + w) / 1e9, "launches": v["n_f"]}
commit = sys.argv[3] if len(sys.argv) > 3 else "unknown"
from gnnome_assembly_amd._lib import DEFAULT_MATMUL_MODE, MATMUL_MODES      # the passes run bench.py as it is: GNM_MATMUL or the library default
mode = os.environ.get("GNM_MATMUL", "").strip().lower() or DEFAULT_MATMUL_MODE
assert mode in MATMUL_MODES, mode
# the whole step: every kernel of the profiled run (bench.py --steps 1 --warmup 1 = warm-up + timed + per-op step), divided by
# the number of steps in it (the loss kernel runs once per step)
steps = max(1, res.get("bce_fwd_bwd_k", {}).get("n_f", 0))
step_total_gb = sum((2.0 * v["fetch_kib"] + v["write_kib"]) * 1024 for v in res.values() if v["n_f"] and v["n_w"]) / steps / 1e9
json.dump({"csrc_sha": csrc_sha(), "steps_profiled": steps, "per_step_total_gb": step_total_gb,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_traffic.sh) on one bench.py "
                     f"step (E=7540278, N=1500000, H=128, L=8, {mode} matmul mode); FETCH_SIZE doubled (gfx950)",
           "commit": commit,
           "workload": {"edges": 7540278, "nodes": 1500000, "hidden": 128, "matmul": mode},   # bench.py defaults (R=750k, seed 0)
           "per_launch": table}, open(out, "w"), indent=1, sort_keys=True)
print(f"whole step: {step_total_gb:.1f} GB ({steps} steps profiled); per layer (1/8 of the kernels that run once per layer) see the table")
for k, v in sorted(table.items(), key=lambda kv: -kv[1]["total_gb"])[:24]:
    print(f"{k:40s} fetch={v['fetch_gb']:7.2f} GB write={v['write_gb']:7.2f} GB total={v['total_gb']:7.2f} GB x {v['launches'] / steps:g} per step")
