cd $GRAFT_REPO_ROOT
LTRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_dstamp.so timeout 200 python tools/lab/mha_dkdv_stamps.py > gpurun_out/r3_dkdv_stamps.txt 2>&1; cat gpurun_out/r3_dkdv_stamps.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or wgrad" 2>&1 | tail -2
bash tools/prof_step.sh r3_slab | grep -i "slab\|tn256\|layernorm"
