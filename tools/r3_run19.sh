cd $GRAFT_REPO_ROOT
LTRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_mhaold.so timeout 900 python -m pytest tests/test_gpu_benchdims.py -x -q -m gpu -k "config4 and neuralNDCG" 2>&1 | tail -3
cp gpurun_out/parity_benchdims_cfg4_neuralNDCG_split_bf16.json gpurun_out/parity_cfg4_neural_oldmha.json
timeout 900 python -m pytest tests/test_gpu_benchdims.py -x -q -m gpu -k "config4 and neuralNDCG" 2>&1 | tail -3
