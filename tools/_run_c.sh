O=gpurun_out/r03c; mkdir -p $O
for v in timing timing_old; do
  GNM_LIBRARY=$GRAFT_REPO_ROOT/tools/chain_phase_timing/libgnm_$v.so python tools/chain_phase_timing/run.py > $O/phase_$v.txt 2>&1; tail -14 $O/phase_$v.txt
done
for cap in "0 4" "1 4" "2 4" "0 3" "0 5" "0 0"; do set -- $cap
  GNM_TN_CAP=$1 GNM_SRC_CAP=$2 python bench.py --steps 10 --warmup 3 --no-alt-orders --no-cpu-baseline --no-alt-matmul > $O/bench_tn$1_src$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_tn$1_src$2.json')); print('tn cap $1 src cap $2:', round(d['ms_per_step'],2))"
done
