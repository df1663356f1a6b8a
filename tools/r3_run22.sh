cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "gemm or fused or trainer or dropout or fit or end_to_end or golden" 2>&1 | tail -5
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
bash tools/prof_step.sh r3_resid | head -12
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-250
