cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2_gputests.log
python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err
bash tools/prof_step.sh r2b > /dev/null 2>&1
bash tools/prof_step.sh r2b_bf16 --gemm bf16 > /dev/null 2>&1
bash tools/lab/pmc_gemm2.sh > /dev/null 2>&1
bash tools/bench_set.sh > gpurun_out/r2_benchset2.log 2>&1
