#!/bin/bash
# Same-box A/B of environment switches on the mini-batch epoch (run on the GPU box via gpurun):
#   tools/ab_minibatch_env.sh "GNM_X=1" "GNM_TN_SIDE=0" ...   -> steady-state M edges/s of tools/minibatch_epoch.py per setting
cd ${GRAFT_REPO_ROOT:-.}
for S in "$@"; do
  env $S python tools/minibatch_epoch.py > /dev/null 2>&1
  python -c "
import json;b=json.load(open('gpurun_out/minibatch.json'));print('$S', round(b['steady_state_edges_per_s']/1e6,2), 'M edges/s')"
done
