O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/gputest.log 2>&1; tail -8 $O/gputest.log
python bench.py --steps 10 --warmup 3 --no-alt-orders --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['value']); 
for k,v in list(d['op_ms'].items())[:14]: print(f'{v:8.3f} {k}')"
