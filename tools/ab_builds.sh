#!/bin/bash
# Same-box A/B of two builds of libgnm.so through GNM_LIBRARY (run on the GPU box via gpurun):
#   tools/ab_builds.sh A.so B.so [rounds] [extra bench.py args]   -> ms/step and the ops above 2.5 ms, alternating A B A B ...
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq $R); do for L in $A $B; do
  GNM_LIBRARY=$L python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders "$@" 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$L'.split('/')[-1], round(b['ms_per_step'],2), {k:round(v,2) for k,v in b['op_ms'].items() if v>2.5})"
done; done
