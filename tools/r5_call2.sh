# round 5, GPU call 2: activation operand images -- bit-identity tests, same-box A/B, per-kernel times, PMC VALU counts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "image or relu_bits or overlap_wgrad_step or one_bit" > gpurun_out/r5c2_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r5c2_pytest.log
timeout 900 python -m pytest tests/test_gpu_benchdims.py -x -q -m gpu -k "bench_batch or 64_slates or config4" > gpurun_out/r5c2_pytest_bd.log 2>&1; echo "benchdims rc=$?"; tail -5 gpurun_out/r5c2_pytest_bd.log
AB_OPT=act_images timeout 600 python tools/wgrad_group_ab.py 64 256 > gpurun_out/r5c2_images_ab.log 2>&1; cat gpurun_out/r5c2_images_ab.log
bash tools/pmc_ab.sh images act_images=1 act_images=0 > gpurun_out/r5c2_pmc.log 2>&1; head -30 gpurun_out/pmc_ab_images.md
bash tools/prof_step.sh r5c2_default > /dev/null 2>&1; head -25 gpurun_out/prof_r5c2_default.txt
