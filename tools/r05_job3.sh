# round 5, GPU job 3: the whole GPU tier after the kernel deletions / new gradient-clause logic, A/B of the deferred weight
# gradient's position, the fp32-MFMA mode on the two-sided backward sweep, kernel profile of a mini-batch epoch
set -x
O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -8 $O/gputest.log
cp gpurun_out/grad_clauses.json gpurun_out/grad_clauses.txt gpurun_out/grad_parity_fullsize.txt $O/ 2>/dev/null
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders"
ab() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name', round(b['ms_per_step'],2), 'ms/step;', {k:round(v,2) for k,v in b['op_ms'].items() if v>3.0})" >> $O/ab_tn_at.txt; }
for i in 1 2 3; do
ab "default (tn012 deferred to the next iteration)" GNM_X=1
ab "TN_AT=now" GNM_TN_AT=now
ab "TN_SIDE=0 (everything on one stream)" GNM_TN_SIDE=0
done
ab "round4 schedule" GNM_NODE_FUSED=0
cat $O/ab_tn_at.txt
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-orders --matmul f32 --no-alt-matmul > $O/bench_f32.json 2>/dev/null
GNM_TWO_SIDED=0 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-orders --matmul f32 --no-alt-matmul > $O/bench_f32_sep.json 2>/dev/null
python -c "
import json
for f in ('bench_f32','bench_f32_sep'):
    b=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(b['ms_per_step'],2), {k:round(v,2) for k,v in b['op_ms'].items() if v>3.0})"
HEAD=60 tools/kernel_stats.sh r05mb python $GRAFT_REPO_ROOT/tools/minibatch_epoch.py --epochs 3 > $O/minibatch_kernel_stats.txt 2>&1
cp $(find gpurun_out/prof_r05mb -name "*kernel_stats.csv" | head -1) $O/minibatch_kernel_stats.csv
HEAD=60 tools/kernel_stats.sh r05s python $GRAFT_REPO_ROOT/bench.py --reads 110000 --steps 10 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/R110k_kernel_stats.txt 2>&1
cp $(find gpurun_out/prof_r05s -name "*kernel_stats.csv" | head -1) $O/R110k_kernel_stats.csv
ls -la $O
