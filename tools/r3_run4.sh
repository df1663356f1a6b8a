# round-3 GPU call 4: full -m gpu suite, smoke, default bench, kernel stats (default + NeuralNDCG workload), PMC of the NeuralNDCG and GEMM kernels, bench set
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3_pytest4.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r3_pytest4.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/r3_smoke4.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_smoke4.log
timeout 400 python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r3_bench_final.json
bash tools/prof_step.sh r3_default > /dev/null 2>&1; head -14 gpurun_out/prof_r3_default.txt
bash tools/prof_step.sh r3_attn_neuralndcg_new --workload attn_neuralndcg > /dev/null 2>&1; head -10 gpurun_out/prof_r3_attn_neuralndcg_new.txt
bash tools/lab/pmc_neural.sh > /dev/null 2>&1; cat gpurun_out/pmc_neural.txt
bash tools/lab/pmc_gemm2.sh > /dev/null 2>&1; cat gpurun_out/pmc_gemm2.txt
timeout 600 bash tools/bench_set.sh > gpurun_out/r3_benchset.log 2>&1; cat gpurun_out/r3_benchset.log
