cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_benchdims_*.json
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -6
