# usage: gpurun -- "bash tools/round_evidence_r3.sh" -- round-3 evidence run: full -m gpu suite, smoke, default bench line, kernel stats per BASELINE workload, PMC of GEMM / attention kernels, bench set (profiles/r03_* come from its gpurun_out/ files)
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_benchdims_*.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3f_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3f_pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r3f_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3f_smoke.log
timeout 500 python bench.py > gpurun_out/r3f_bench_default.json 2> gpurun_out/r3f_bench_default.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r3f_bench_default.json
bash tools/prof_step.sh r3f_default > /dev/null 2>&1; head -16 gpurun_out/prof_r3f_default.txt
for w in attn_neuralndcg attn_lambdarank attn1024_listmle fc_listnet; do bash tools/prof_step.sh r3f_$w --workload $w > /dev/null 2>&1; echo "== $w"; head -8 gpurun_out/prof_r3f_$w.txt; done
bash tools/lab/pmc_mha.sh > /dev/null 2>&1; cat gpurun_out/pmc_mha.txt
bash tools/lab/pmc_gemm2.sh > /dev/null 2>&1; cat gpurun_out/pmc_gemm2.txt
timeout 700 bash tools/bench_set.sh > gpurun_out/r3f_benchset.log 2>&1; cat gpurun_out/r3f_benchset.log
