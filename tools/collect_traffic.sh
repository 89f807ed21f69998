#!/bin/bash
# HBM traffic per kernel launch from the TCC PMC counters (run on the GPU box via gpurun).
# Two separate rocprofv3 passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit
# together), kernel-trace only, as MI355X_MICROARCH.md prescribes, over ONE bench.py step (encoders, 8 layers with
# the chained backward, predictor) -- or over tools/microbench.py (one un-chained layer) with LAYER=1.
# Output: gpurun_out/traffic/*.csv ; summarise with tools/traffic_summary.py
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/${OUTDIR:-traffic}   # OUTDIR: directory under gpurun_out/; EXTRA: further bench.py arguments (e.g. --shuffle-nodes)
mkdir -p $D
cd /tmp
if [ -n "$LAYER" ]; then CMD="python $R/tools/microbench.py --iters 2 --matmul bf16x3"; else CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-matmul --no-alt-orders $EXTRA"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o $c -- $CMD > $D/$c.log 2>&1
done
ls $D
