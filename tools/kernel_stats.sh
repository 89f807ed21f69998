#!/bin/bash
# rocprofv3 --kernel-trace --stats of a command, on the GPU box: tools/kernel_stats.sh NAME cmd...
# Writes gpurun_out/prof_NAME/*kernel_stats.csv and prints its head.  Every step is bounded.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
out=$R/gpurun_out/prof_$name
mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- "$@" > $out/cmd.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -${HEAD:-25} "$f" | cut -c1-220; else echo "no kernel_stats.csv"; tail -5 $out/cmd.log; fi
find $out -name "*kernel_trace.csv" -delete   # large
