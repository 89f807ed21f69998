# round 5, GPU job 5: the mini-batch mode with and without the prefetching loader; the tests that touch it
set -x
O=gpurun_out/r05e; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "minibatch or training_harness or train_loop or bench_line" > $O/gputest_mb.log 2>&1; tail -5 $O/gputest_mb.log
python tools/minibatch_epoch.py > $O/minibatch_prefetch.log 2>&1; tail -1 $O/minibatch_prefetch.log | cut -c1-600; cp gpurun_out/minibatch.json $O/minibatch_prefetch.json
GNM_BATCH_PREFETCH=0 python tools/minibatch_epoch.py > $O/minibatch_noprefetch.log 2>&1; tail -1 $O/minibatch_noprefetch.log | cut -c1-600
GNM_DEVICE_PLANS=0 python tools/minibatch_epoch.py > $O/minibatch_prefetch_noplans.log 2>&1; tail -1 $O/minibatch_prefetch_noplans.log | cut -c1-600
python tools/minibatch_breakdown.py > $O/minibatch_breakdown.txt 2>&1; cat $O/minibatch_breakdown.txt
python tools/minibatch_epoch.py --shuffle-nodes > $O/minibatch_shuffled.log 2>&1; tail -1 $O/minibatch_shuffled.log | cut -c1-600
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/bench_tags.json 2> $O/bench_tags.err; python -c "
import json; b=json.loads(open('$O/bench_tags.json').read().strip().splitlines()[-1]); print(round(b['ms_per_step'],2), {k:round(v,2) for k,v in b['op_ms'].items() if v>1.5}); print(b['roofline'].get('traffic_source'))"
