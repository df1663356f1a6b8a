cd $GRAFT_REPO_ROOT
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | cut -c1-400
bash tools/lab/pmc_neural.sh > /dev/null 2>&1; cat gpurun_out/pmc_neural.txt
bash tools/lab/pmc_ln.sh > /dev/null 2>&1; cat gpurun_out/pmc_ln.txt 2>/dev/null | head -40
