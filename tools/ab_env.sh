#!/bin/bash
# Same-box A/B of environment switches (run on the GPU box via gpurun):
#   tools/ab_env.sh "GNM_TN_SPLIT=0" "GNM_TN_SPLIT=1" [rounds] [extra bench.py args]   -> ms/step, alternating A B A B ...
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do for S in "$A" "$B"; do
  env $S python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders "$@" 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$S', round(b['ms_per_step'],2))"
done; done
