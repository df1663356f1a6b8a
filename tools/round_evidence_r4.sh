# usage: gpurun -- "bash tools/round_evidence_r4.sh [quick]" -- round-4 evidence: the full bench JSON line of EVERY BASELINE workload
# (gpurun_out/r4_bench_<workload>_line.json -> profiles/r04_bench_<workload>_line.json), rocprofv3 kernel stats per workload, the
# fc_listnet PMC passes, the bench set
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in attn_approxndcg fc_listnet attn_neuralndcg attn_lambdarank attn1024_listmle; do
  extra=""; [ "$w" != "attn_approxndcg" ] && [ "$w" != "fc_listnet" ] && extra="--no-cpu-baseline"
  timeout 600 python bench.py --workload $w $extra > gpurun_out/r4_bench_${w}_line.json 2> gpurun_out/r4_bench_${w}.err; echo "bench $w rc=$?"; cut -c1-260 gpurun_out/r4_bench_${w}_line.json
done
for w in attn_approxndcg fc_listnet attn_neuralndcg attn_lambdarank attn1024_listmle; do bash tools/prof_step.sh r4_$w --workload $w > /dev/null 2>&1; echo "== $w"; head -8 gpurun_out/prof_r4_$w.txt; done
bash tools/prof_step.sh r4_fc_listnet_b2048 --workload fc_listnet --slates-per-gpu 2048 > /dev/null 2>&1; echo "== fc_listnet 2048"; head -6 gpurun_out/prof_r4_fc_listnet_b2048.txt
bash tools/prof_step.sh r4_attn_b64 --slates-per-gpu 64 > /dev/null 2>&1; echo "== attn 64 slates"; head -14 gpurun_out/prof_r4_attn_b64.txt
[ "$1" = "quick" ] || { timeout 700 bash tools/bench_set.sh > gpurun_out/r4_benchset.log 2>&1; cat gpurun_out/r4_benchset.log; }
