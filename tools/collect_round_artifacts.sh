# Round artifacts on the GPU box (run through gpurun): everything that is copied into profiles/ afterwards.
set -x
T=${TAG:-r03}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "test_model_matches_golden" 2>&1 | grep "logits rel_l2" > $O/logit_parity.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
HEAD=45 tools/kernel_stats.sh $T python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/ks.txt 2>&1
cp $(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
OUTDIR=traffic tools/collect_traffic.sh > $O/traffic.log 2>&1
python tools/traffic_summary.py gpurun_out/traffic $O/traffic.json $(cat .git_head) > $O/traffic.txt 2>&1
OUTDIR=traffic_shuf EXTRA="--shuffle-nodes" tools/collect_traffic.sh > $O/traffic_shuf.log 2>&1
python tools/traffic_summary.py gpurun_out/traffic_shuf $O/traffic_shuffled.json $(cat .git_head) > $O/traffic_shuffled.txt 2>&1
OUTDIR=traffic_shuf_kept EXTRA="--shuffle-nodes --node-order keep" tools/collect_traffic.sh > $O/traffic_shuf_kept.log 2>&1
python tools/traffic_summary.py gpurun_out/traffic_shuf_kept $O/traffic_shuffled_kept.json $(cat .git_head) > $O/traffic_shuffled_kept.txt 2>&1
BENCH=1 tools/pmc_run.sh $T --no-alt-orders > $O/sq_pmc_bench.txt 2>&1
for R in 110000 500000 1000000; do python bench.py --reads $R --steps 5 --warmup 2 --no-cpu-baseline --no-alt-orders > $O/train_R$R.json 2>/dev/null; done
for R in 750000 3000000; do python bench.py --reads $R --inference --steps 5 --warmup 2 --no-cpu-baseline > $O/infer_R$R.json 2>/dev/null; done
python bench.py --hidden 256 --reads 375000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul --no-alt-orders > $O/h256.json 2>/dev/null
python bench.py --shuffle-nodes --steps 10 --warmup 3 --no-cpu-baseline --no-alt-matmul > $O/bench_shuffled_nodes.json 2>/dev/null
ls -la $O
