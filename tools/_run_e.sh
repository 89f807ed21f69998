O=gpurun_out/r03e; mkdir -p $O
CHAIN3=1 GNM_LIBRARY=$GRAFT_REPO_ROOT/tools/chain_phase_timing/libgnm_timing3.so python tools/chain_phase_timing/run.py > $O/phase_chain3.txt 2>&1; tail -8 $O/phase_chain3.txt
