O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -25 $O/gputest.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d.get('alt_node_order'),indent=1)); print(json.dumps(d.get('alt_edge_ids'))); print(d['cpu_baseline'])"
