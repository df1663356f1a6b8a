cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_benchdims.py -x -q -m gpu -k "config4" 2>&1 | tail -3
