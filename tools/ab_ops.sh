#!/bin/bash
# Same-box A/B of environment switches with the per-op times of chosen ops beside the step (run on the GPU box via gpurun):
#   OPS="gnm_edge_gate2_fwd gnm_edge_t_fused_fwd" tools/ab_ops.sh "ENV_A=.." "ENV_B=.." [rounds] [extra bench.py args]
# (bench.py's op_ms comes from a serialised, event-bracketed step; ms/step from the free-running timed steps)
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq $R); do for S in "$A" "$B"; do
  env $S OPS="$OPS" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders "$@" 2>/dev/null | python -c "
import json,sys,os;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);ops=os.environ.get('OPS','').split()
print('$S', round(b['ms_per_step'],2), 'ms/step;', {k:round(v,2) for k,v in b.get('op_ms',{}).items() if (k in ops if ops else v>3.0)}, 'peak GiB', b.get('peak_mem_gib'))"
done; done
