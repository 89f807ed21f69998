O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for cap in "0 4" "1 4" "0 4" "1 4" "1 3" "1 5"; do set -- $cap
  GNM_TN_CAP=$1 GNM_SRC_CAP=$2 python bench.py --steps 10 --warmup 3 --no-alt-orders --no-cpu-baseline --no-alt-matmul > $O/bench_tn$1_src$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_tn$1_src$2.json')); print('tn cap $1 src cap $2:', round(d['ms_per_step'],2))"
done
