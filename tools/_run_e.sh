O=gpurun_out/r03e; mkdir -p $O
for v in old new; do
GNM_LIBRARY=$GRAFT_REPO_ROOT/tools/chain_phase_timing/libgnm_timing_$v.so python tools/chain_phase_timing/run.py > $O/phase_$v.txt 2>&1; tail -13 $O/phase_$v.txt
done
