#!/bin/bash
# Builds scratch_abl/libgnm_timing.so = the current sources + per-phase s_memtime stamps in edge_bwd_chain_k (patch.py);
# run on the GPU box with  GNM_LIBRARY=$GRAFT_REPO_ROOT/scratch_abl/libgnm_timing.so python tools/chain_phase_timing/run.py
# (cycles per tile and phase, per wave).  Needs build/*.o of a normal build (python __graft_entry__.py).
set -e
cd /root/repo
rm -rf scratch_abl/csrc_t; mkdir -p scratch_abl/csrc_t scratch_abl/obj; cp gnnome_assembly_amd/csrc/* scratch_abl/csrc_t/
python3 tools/chain_phase_timing/patch.py
cd scratch_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I /root/repo/include -I csrc_t -c csrc_t/gnm_tr.hip -o obj/gnm_tr_t.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgnm_timing.so obj/gnm_tr_t.o $(ls /root/repo/build/*.o | grep -v gnm_tr.o)
