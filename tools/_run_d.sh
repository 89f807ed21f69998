O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "chained or lean or golden or branch or chr19 or side_stream" > $O/gputest.log 2>&1; tail -6 $O/gputest.log
for v in 0 0; do
  GNM_VARIANTS=chain=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-alt-orders --no-cpu-baseline --no-alt-matmul > $O/bench_chain$v.json 2>$O/bench_chain$v.err
  python -c "
import json; d=json.load(open('$O/bench_chain$v.json')); print('chain variant $v:', round(d['ms_per_step'],2), 'chain op ms', d['op_ms'].get('gnm_edge_bwd_chain'))"
done
