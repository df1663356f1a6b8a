cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "test_fit_has_the_reference_contract or test_end_to_end_training_on_device or large_tile_gemm or short_last_batch" > gpurun_out/r3_pytest2.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest2.log
timeout 600 python tools/gemm_persist_ab.py > gpurun_out/r3_gemm_persist_ab_256.txt 2>&1; cat gpurun_out/r3_gemm_persist_ab_256.txt
GSLATES=64 timeout 600 python tools/gemm_persist_ab.py > gpurun_out/r3_gemm_persist_ab_64.txt 2>&1; cat gpurun_out/r3_gemm_persist_ab_64.txt
