#!/bin/bash
# SQ counter pass over an arbitrary command on the GPU box: tools/pmc_cmd.sh NAME cmd...   (absolute paths: the
# command runs from /tmp).  kernel-trace only, as MI355X_MICROARCH.md prescribes.  Summary -> gpurun_out/pmc_NAME.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
out=$R/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --kernel-trace --output-format csv -d $out -o pmc -- "$@" > $out/cmd.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python $R/tools/pmc_summary.py "$f" | tee $R/gpurun_out/pmc_$name.txt; rm -f "$f"; else echo "no counters"; tail -5 $out/cmd.log; fi
find $out -name "*kernel_trace.csv" -delete
