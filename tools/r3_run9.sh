cd $GRAFT_REPO_ROOT
LTRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_nstamp.so timeout 200 python tools/lab/neural_stamps.py > gpurun_out/r3_neural_stamps2.txt 2>&1; cat gpurun_out/r3_neural_stamps2.txt
timeout 200 python tools/neural_ab.py 2>&1 | grep "^{"
