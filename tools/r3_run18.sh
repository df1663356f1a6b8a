cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/r3_bench_ds.json; cut -c1-330 gpurun_out/r3_bench_ds.json
