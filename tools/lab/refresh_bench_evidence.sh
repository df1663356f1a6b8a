cd $GRAFT_REPO_ROOT
timeout 500 python bench.py > gpurun_out/r3f_bench_default.json 2> gpurun_out/r3f_bench_default.err; echo "bench rc=$?"; cut -c1-220 gpurun_out/r3f_bench_default.json
bash tools/prof_step.sh r3f_default > /dev/null 2>&1; head -12 gpurun_out/prof_r3f_default.txt
for w in attn_neuralndcg attn_lambdarank attn1024_listmle fc_listnet; do bash tools/prof_step.sh r3f_$w --workload $w > /dev/null 2>&1; done
timeout 700 bash tools/bench_set.sh > gpurun_out/r3f_benchset.log 2>&1; cat gpurun_out/r3f_benchset.log
