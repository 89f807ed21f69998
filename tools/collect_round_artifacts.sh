# Round artifacts on the GPU box (run through gpurun): everything that is copied into profiles/ afterwards.
#   TAG=r06 tools/collect_round_artifacts.sh          (FAST=1: skip the full GPU test tier and the sweeps;
#                                                      PART=1 / PART=2: the first / second half only -- a gpurun call is limited to an hour)
set -x
T=${TAG:-r06}; O=gpurun_out/$T; mkdir -p $O
if [ "$PART" != "2" ]; then
if [ -z "$FAST" ]; then
  python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -3 $O/gputest.log
  cp gpurun_out/logit_parity_fullsize.txt gpurun_out/grad_parity_fullsize.txt gpurun_out/rccl_smoke.log gpurun_out/grad_clauses.json gpurun_out/grad_clauses.txt gpurun_out/cabi_host_step.txt gpurun_out/dp2_train_config4.log gpurun_out/accuracy_e2e.txt $O/ 2>/dev/null
  python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "test_model_matches_golden" 2>&1 | grep "logits rel_l2" > $O/logit_parity.txt
fi
HEAD=45 tools/kernel_stats.sh $T python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/ks.txt 2>&1
cp $(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
# round 6: the same table with everything on ONE stream (GNM_TN_SIDE=0): kernel durations then add up to the step -- in the two-stream
# trace a kernel that waits for CUs beside a side-stream kernel shows the wait as its own duration (slab_reduce_k: 1.4 ms "average"
# against 73 us alone)
GNM_TN_SIDE=0 HEAD=45 tools/kernel_stats.sh ${T}_serial python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/ks_serial.txt 2>&1
cp $(find gpurun_out/prof_${T}_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
OUTDIR=traffic tools/collect_traffic.sh > $O/traffic.log 2>&1
python tools/traffic_summary.py gpurun_out/traffic $O/traffic.json $(cat .git_head) > $O/traffic.txt 2>&1
# the bench line is taken AFTER the PMC pass so that its roofline.traffic is this tree's own (bench.py checks the csrc hash)
cp $O/traffic.json profiles/${T}_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
BENCH=1 tools/pmc_run.sh $T --no-alt-orders > $O/sq_pmc_bench.txt 2>&1
python tools/minibatch_epoch.py > $O/minibatch.log 2>&1; cp gpurun_out/minibatch.json $O/ 2>/dev/null
python tools/minibatch_breakdown.py > $O/minibatch_breakdown.txt 2>&1
fi
if [ -z "$FAST" ] && [ "$PART" != "1" ]; then
  OUTDIR=traffic_shuf EXTRA="--shuffle-nodes" tools/collect_traffic.sh > $O/traffic_shuf.log 2>&1
  python tools/traffic_summary.py gpurun_out/traffic_shuf $O/traffic_shuffled.json $(cat .git_head) > $O/traffic_shuffled.txt 2>&1
  for R in 110000 375000 500000 1000000; do python bench.py --reads $R --steps 5 --warmup 2 --no-cpu-baseline --no-alt-orders > $O/train_R$R.json 2>/dev/null; done
  for R in 750000 3000000; do python bench.py --reads $R --inference --steps 5 --warmup 2 --no-cpu-baseline > $O/infer_R$R.json 2>/dev/null; done
  python bench.py --hidden 256 --reads 375000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h256.json 2>/dev/null
  python bench.py --hidden 256 --layers 16 --reads 110000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h256_l16.json 2>/dev/null
  python bench.py --shuffle-nodes --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul > $O/bench_shuffled_nodes.json 2>/dev/null
  python tools/minibatch_epoch.py --shuffle-nodes > $O/minibatch_shuffled.log 2>&1; cp gpurun_out/minibatch_shuffled.json $O/ 2>/dev/null
  # same-box A/B of the round-5 node-side switches against the round-4 schedule (three alternating runs)
  B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders"
  ab() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name', round(b['ms_per_step'],2), 'ms/step;', {k:round(v,2) for k,v in b['op_ms'].items() if v>3.0})" >> $O/ab_node_side_final.txt; }
  for i in 1 2 3; do
    ab "round 5 (f16x2, NODE_FUSED, TN_AT=now)" GNM_X=1
    ab "bf16x3 matmul mode" GNM_MATMUL=bf16x3
    ab "round-4 schedule and matmul mode (bf16x3, NODE_FUSED=0, TN_AT=next)" GNM_MATMUL=bf16x3 GNM_NODE_FUSED=0 GNM_TN_AT=next
  done
  python tools/matmul_accuracy.py > $O/f16x2_accuracy.log 2>&1; cp gpurun_out/f16x2_accuracy.txt $O/ 2>/dev/null
  # round 6: 64-wide models zero-padded to 128 against the native 64-wide route; the forward sweep at one workgroup per CU; LayerNorm per op; a wide model
  { for LL in 1 8; do echo "H = 64, L = $LL"; tools/ab_env.sh GNM_NATIVE_64=1 GNM_NATIVE_64=0 2 --hidden 64 --layers $LL; done; } > $O/h64_padded.txt 2>&1
  OPS="gnm_edge_gate2_fwd gnm_edge_t_fused_fwd" tools/ab_ops.sh "GNM_GATE2_WG=2" "GNM_GATE2_WG=1 GNM_VARIANTS=gate2_wg=1" 2 > $O/ab_gate2_occupancy.txt 2>&1
  python tools/ln_two_sided_ab.py > /dev/null 2>&1; cp gpurun_out/ln_two_sided.txt $O/ 2>/dev/null
  # the reference's default configuration end to end: mini-batch mode (500 parts, 50 per batch), dim_latent 256, num_gnn_layers 16, the true chr19 size
  python tools/minibatch_epoch.py --hidden 256 --layers 16 --reads 110000 --epochs 4 > $O/minibatch_h256l16.log 2>&1; cp gpurun_out/minibatch_h256l16.json $O/ 2>/dev/null
  python bench.py --hidden 512 --layers 2 --reads 200000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h512_l2_R200k.json 2>/dev/null
  # the schedule by graph size (engine.TN_AT = auto) and kernel tables of the other shapes
  for R in 110000 375000 750000; do echo "R=$R"; tools/ab_env.sh GNM_TN_AT=now GNM_TN_AT=next 2 --reads $R; done > $O/ab_tn_at_sizes.txt 2>&1
  tools/ab_minibatch_env.sh GNM_TN_AT=now GNM_TN_AT=next GNM_TN_AT=auto >> $O/ab_tn_at_sizes.txt 2>&1
  HEAD=45 tools/kernel_stats.sh h256 python $GRAFT_REPO_ROOT/bench.py --hidden 256 --reads 375000 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-matmul --no-alt-orders > /dev/null 2>&1
  cp $(find gpurun_out/prof_h256 -name "*kernel_stats.csv" | head -1) $O/h256_kernel_stats.csv
  HEAD=45 tools/kernel_stats.sh r110k python $GRAFT_REPO_ROOT/bench.py --reads 110000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > /dev/null 2>&1
  cp $(find gpurun_out/prof_r110k -name "*kernel_stats.csv" | head -1) $O/r110k_kernel_stats.csv
  python tools/power_per_op.py > $O/power_per_op.log 2>&1; cp gpurun_out/power_per_op.txt $O/ 2>/dev/null
  rm -f gpurun_out/power_probe.txt
  python tools/power_probe.py train 10 > /dev/null 2>&1; python tools/power_probe.py forward 6 > /dev/null 2>&1
  GNM_MATMUL=bf16x3 python tools/power_probe.py train 8 > /dev/null 2>&1; GNM_MATMUL=f32 python tools/power_probe.py train 8 > /dev/null 2>&1
  cp gpurun_out/power_probe.txt $O/ 2>/dev/null
fi
ls -la $O
