# round 5, GPU job 4: GPU tier (new clause logic, C++ hosts, slab reduction), bench + kernel stats after the slab-reduction rewrite
set -x
O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -8 $O/gputest.log
cp gpurun_out/grad_clauses.json gpurun_out/grad_clauses.txt gpurun_out/grad_parity_fullsize.txt gpurun_out/cabi_host_step.txt $O/ 2>/dev/null
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders"
ab() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name', round(b['ms_per_step'],2), 'ms/step;', {k:round(v,2) for k,v in b['op_ms'].items() if v>3.0})" >> $O/ab.txt; }
for i in 1 2 3; do
ab "round5 default" GNM_X=1
ab "round4 schedule (NODE_FUSED=0, TN_AT=next)" GNM_NODE_FUSED=0 GNM_TN_AT=next
done
cat $O/ab.txt
HEAD=40 tools/kernel_stats.sh r05d python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/ks.txt 2>&1
cp $(find gpurun_out/prof_r05d -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep -i "slab_reduce\|tn_tr_k\|rowtile_nn" $O/kernel_stats.csv | cut -c1-200
python tools/minibatch_epoch.py > $O/minibatch.log 2>&1; tail -3 $O/minibatch.log; cp gpurun_out/minibatch.json $O/ 2>/dev/null
python bench.py --reads 110000 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/train_R110000.json 2>/dev/null
python bench.py --hidden 256 --reads 375000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h256.json 2>/dev/null
python -c "
import json
for f in ('train_R110000','h256'):
    b=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(b['ms_per_step'],2), round(b['value']/1e6,2))"
ls -la $O
