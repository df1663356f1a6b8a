cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or mha or varlen or compact or reentrant or encoder" 2>&1 | tail -2
bash tools/lab/kern_ab.sh "python tools/mha_one.py" mhaold main > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_mha_ds_kern6.txt; grep -v fwd gpurun_out/r3_mha_ds_kern6.txt
bash tools/lab/kern_ab.sh "ML=1024 MB=64 python tools/mha_one.py" mhaold main > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_mha_ds_kern1024.txt; grep -v fwd gpurun_out/r3_mha_ds_kern1024.txt
bash tools/r3_run15.sh 2>&1 | grep "^== m\|^[0-9]\|mha"
