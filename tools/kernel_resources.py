#!/usr/bin/env python3
"""Print VGPR / AGPR / SGPR / spill / LDS / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "gnnome_assembly_amd", "csrc")
files = sys.argv[1:] or [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
                          "-I", os.path.join(REPO, "include"), "-I", CSRC, "-c", os.path.join(CSRC, f), "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k.startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void gnm::", "")
            print(f"{f:15s} {name:40s} vgpr={cur.get('VGPRs','?'):>3s} agpr={cur.get('AGPRs','?'):>3s} sgpr={cur.get('TotalSGPRs','?'):>3s} "
                  f"spill={cur.get('VGPRs Spill','?')} scratch={cur.get('ScratchSize [bytes/lane]','?')} occ={cur.get('Occupancy [waves/SIMD]','?')} lds={v}")
