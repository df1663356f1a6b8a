cd $GRAFT_REPO_ROOT
LTRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_nstamp.so timeout 200 python tools/lab/neural_stamps.py > gpurun_out/r3_neural_stamps.txt 2>&1; cat gpurun_out/r3_neural_stamps.txt
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "test_fit_ or fused_optimizers or falls_back" > gpurun_out/r3_pytest8.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_pytest8.log
