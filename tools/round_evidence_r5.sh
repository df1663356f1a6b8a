# usage: gpurun -- "bash tools/round_evidence_r5.sh" -- round-5 evidence on ONE box: the full GPU test run, the parity table, the bench JSON
# line of every BASELINE workload (gpurun_out/r5_bench_<workload>_line.json -> profiles/r05_bench_<workload>_line.json), rocprofv3 kernel
# stats per workload, the bench set
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_benchdims_*.json
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r5_gpu_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5_gpu_pytest.log
python tools/parity_table.py gpurun_out/r5_parity_benchdims.md
for w in attn_approxndcg fc_listnet attn_neuralndcg attn_lambdarank attn1024_listmle; do
  extra=""; [ "$w" != "attn_approxndcg" ] && [ "$w" != "fc_listnet" ] && extra="--no-cpu-baseline"
  timeout 600 python bench.py --workload $w $extra > gpurun_out/r5_bench_${w}_line.json 2> gpurun_out/r5_bench_${w}.err; echo "bench $w rc=$?"; cut -c1-200 gpurun_out/r5_bench_${w}_line.json
done
for w in attn_approxndcg fc_listnet attn_neuralndcg attn_lambdarank attn1024_listmle; do bash tools/prof_step.sh r5_$w --workload $w > /dev/null 2>&1; echo "== $w"; head -6 gpurun_out/prof_r5_$w.txt; done
timeout 700 bash tools/bench_set.sh > gpurun_out/r5_benchset.log 2>&1; cat gpurun_out/r5_benchset.log
