# round-3 GPU call 1: full -m gpu suite of the re-entrant ABI tree, smoke, baseline bench, kernel stats of the other BASELINE configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3_pytest1.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r3_pytest1.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r3_smoke1.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r3_smoke1.log
timeout 400 python bench.py > gpurun_out/r3_bench_base.json 2> gpurun_out/r3_bench_base.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r3_bench_base.json
for w in attn_neuralndcg attn_lambdarank attn1024_listmle fc_listnet; do
  bash tools/prof_step.sh r3_$w --workload $w > /dev/null 2>&1
  head -12 gpurun_out/prof_r3_$w.txt
done
