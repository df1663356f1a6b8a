cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or mha or varlen or compact or reentrant or encoder" 2>&1 | tail -4
bash tools/lab/kern_ab.sh "python tools/mha_one.py" mhaold main > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_mha_ds_kern4.txt; cat gpurun_out/r3_mha_ds_kern4.txt
bash tools/lab/kern_ab.sh "ML=1024 MB=64 python tools/mha_one.py" mhaold main > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_mha_ds_kern1024.txt; cat gpurun_out/r3_mha_ds_kern1024.txt
timeout 300 python bench.py 2>&1 | tail -1 | cut -c1-300
