#!/bin/bash
# kernel-trace of tools/microbench.py; prints start/end (us, relative) of the backward kernels of the last iteration
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/trace_overlap
mkdir -p $out; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o tr -- python $R/tools/microbench.py --iters 1 "$@" > $out/cmd.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "gnm::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "edge_bwd_dst" in r["Kernel_Name"]][-1]
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:last + 14]:
    print(f'{r["Kernel_Name"].split("(")[0][-40:]:40s} start {(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  end {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} us  grid {r.get("Grid_Size", "?")}')
PY
rm -f "$f"
