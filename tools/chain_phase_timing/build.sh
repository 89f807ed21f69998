#!/bin/bash
# Builds tools/chain_phase_timing/libgnm_timing.so = the current sources + per-phase clock stamps in edge_bwd_chain_k
# (patch.py); run on the GPU box with
#   GNM_LIBRARY=$GRAFT_REPO_ROOT/tools/chain_phase_timing/libgnm_timing.so python tools/chain_phase_timing/run.py
# (cycles per tile and phase, per wave).  Needs build/*.o of a normal build (python __graft_entry__.py).
set -e
cd /root/repo
W=build/timing
rm -rf $W; mkdir -p $W/csrc_t; cp gnnome_assembly_amd/csrc/* $W/csrc_t/
python3 tools/chain_phase_timing/patch.py $W/csrc_t/gnm_tr.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I /root/repo/include -I $W/csrc_t -c $W/csrc_t/gnm_tr.hip -o $W/gnm_tr_t.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/chain_phase_timing/libgnm_timing.so $W/gnm_tr_t.o $(ls /root/repo/build/*.o | grep -v "build/gnm_tr.o")
ls -la tools/chain_phase_timing/libgnm_timing.so
