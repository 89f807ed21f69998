# round 5, GPU job 2: parity of the node-side kernels / schedules, then a same-box A/B of the switches
set -x
O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "presplit or fused_node or round5 or lean_activation or chained_backward or side_stream or flat_gradient or odd_hidden or test_model_matches_golden" > $O/gputest_new.log 2>&1; tail -15 $O/gputest_new.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "full_size_gradients" > $O/gputest_fullgrad.log 2>&1; tail -8 $O/gputest_fullgrad.log
cp gpurun_out/grad_parity_fullsize.txt $O/ 2>/dev/null
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders"
ab() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name', round(b['ms_per_step'],2), 'ms/step;', {k:round(v,2) for k,v in b['op_ms'].items() if v>2.0})" >> $O/ab_node_side.txt; }
ab "round4 (PRESPLIT=0 NODE_FUSED=0)" GNM_PRESPLIT=0 GNM_NODE_FUSED=0
ab "round5 default" GNM_X=1
ab "PRESPLIT only" GNM_NODE_FUSED=0
ab "NODE_FUSED only" GNM_PRESPLIT=0
ab "default, layer 0 on fp32 operands" GNM_PRESPLIT_L0=0
ab "default, tn012 right away (TN_AT=now)" GNM_TN_AT=now
ab "default, pre-split tn at 3 WG/CU" GNM_VARIANTS=tn_s3_occ=3
ab "round4 again" GNM_PRESPLIT=0 GNM_NODE_FUSED=0
ab "round5 default again" GNM_X=1
cat $O/ab_node_side.txt
