#!/bin/bash
# HBM traffic per kernel launch from the TCC PMC counters (run on the GPU box via gpurun).
# Two separate rocprofv3 passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit
# together), kernel-trace only, as MI355X_MICROARCH.md prescribes, over ONE bench.py step (encoders, 8 layers with
# the chained backward, predictor) -- or over tools/microbench.py (one un-chained layer) with LAYER=1.
# Output: gpurun_out/traffic/*.csv ; summarise with tools/traffic_summary.py
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/traffic
cd /tmp
if [ -n "$LAYER" ]; then CMD="python $R/tools/microbench.py --iters 2 --matmul bf16x3"; else CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-matmul"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic -o $c -- $CMD > $R/gpurun_out/traffic/$c.log 2>&1
done
ls $R/gpurun_out/traffic
