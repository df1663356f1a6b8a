# usage: gpurun -- "bash tools/round_evidence_r6.sh [suite|bench|all]" -- round-6 evidence on ONE box: the full GPU test run (with durations), the
# parity table, the bench JSON line of every BASELINE workload (gpurun_out/r6_bench_<workload>_line.json -> profiles/r06_bench_<workload>_line.json),
# the one-rank RCCL line (--force-dist), rocprofv3 kernel stats per workload, the bench set
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
what=${1:-all}
mkdir -p gpurun_out
if [ "$what" = suite ] || [ "$what" = all ]; then
  rm -f gpurun_out/parity_benchdims_*.json
  timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r6_gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -32 gpurun_out/r6_gpu_pytest.log
  python tools/parity_table.py gpurun_out/r6_parity_benchdims.md
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
  for w in attn_approxndcg fc_listnet attn_neuralndcg attn_lambdarank attn1024_listmle; do
    extra=""; [ "$w" != "attn_approxndcg" ] && [ "$w" != "fc_listnet" ] && extra="--no-cpu-baseline"
    timeout 900 python bench.py --workload $w $extra > gpurun_out/r6_bench_${w}_line.json 2> gpurun_out/r6_bench_${w}.err; echo "bench $w rc=$?"; cut -c1-200 gpurun_out/r6_bench_${w}_line.json
  done
  timeout 600 python bench.py --force-dist --no-cpu-baseline --no-side-pass > gpurun_out/r6_bench_force_dist_line.json 2> gpurun_out/r6_bench_force_dist.err; echo "force-dist rc=$?"; cut -c1-200 gpurun_out/r6_bench_force_dist_line.json
  for w in attn_approxndcg fc_listnet attn_neuralndcg attn_lambdarank attn1024_listmle; do bash tools/prof_step.sh r6_$w --workload $w > /dev/null 2>&1; echo "== $w"; head -6 gpurun_out/prof_r6_$w.txt; done
  timeout 700 bash tools/bench_set.sh > gpurun_out/r6_benchset.log 2>&1; cat gpurun_out/r6_benchset.log
fi
