set -x
O=gpurun_out/r02i; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "test_model_matches_golden" 2>&1 | grep "logits rel_l2" > $O/logit_parity.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
HEAD=40 tools/kernel_stats.sh r02i python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-matmul > $O/ks.txt 2>&1
cp $(find gpurun_out/prof_r02i -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
tools/collect_traffic.sh > $O/traffic.log 2>&1
python tools/traffic_summary.py gpurun_out/traffic $O/traffic.json $(cat .git_head) > $O/traffic.txt 2>&1
BENCH=1 tools/pmc_run.sh r02i > $O/sq_pmc_bench.txt 2>&1
for R in 110000 500000 1000000; do python bench.py --reads $R --steps 5 --warmup 2 --no-cpu-baseline > $O/train_R$R.json 2>/dev/null; done
for R in 750000 3000000; do python bench.py --reads $R --inference --steps 5 --warmup 2 --no-cpu-baseline > $O/infer_R$R.json 2>/dev/null; done
python bench.py --hidden 256 --reads 375000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-matmul > $O/h256.json 2>/dev/null
ls -la $O
