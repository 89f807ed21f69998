# round 5, GPU job 1: baseline numbers of the round-4 tree on this round's box + the full-size fp64 gradient fixture (host CPU work)
set -x
O=gpurun_out/r05a; mkdir -p $O
nproc; free -g | head -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/base_bench.json 2> $O/base_bench.err
python bench.py --hidden 256 --layers 16 --reads 110000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h256_l16_R110k.json 2> $O/h256_l16_R110k.err
GNM_ACTIVATIONS=lean python bench.py --hidden 256 --layers 16 --reads 375000 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h256_l16_R375k_lean.json 2> $O/h256_l16_R375k_lean.err
python tools/minibatch_breakdown.py > $O/minibatch_breakdown_base.txt 2>&1
# the fp64 oracle's gradients at the metric's size (20-30 min of host CPU); the GPU test tier runs beside it
( python tests/golden/make_golden_fullsize.py --grads > $O/make_grads.log 2>&1; cp tests/golden/fullsize_grads_r750k.npz $O/ ) &
GP=$!
python -m pytest tests -q -m gpu -x > $O/gputest_base.log 2>&1; tail -3 $O/gputest_base.log
wait $GP
tail -3 $O/make_grads.log; ls -la $O
