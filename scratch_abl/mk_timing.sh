#!/bin/bash
# builds scratch_abl/libgnm_timing.so = current sources + per-phase s_memtime stamps in the chain kernel
set -e
cd /root/repo
rm -rf scratch_abl/csrc_t; mkdir -p scratch_abl/csrc_t scratch_abl/obj; cp gnnome_assembly_amd/csrc/* scratch_abl/csrc_t/
python3 scratch_abl/patch_timing.py
cd scratch_abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I /root/repo/include -I csrc_t -c csrc_t/gnm_tr.hip -o obj/gnm_tr_t.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgnm_timing.so obj/gnm_tr_t.o $(ls /root/repo/build/*.o | grep -v gnm_tr.o)
