cd $GRAFT_REPO_ROOT
timeout 300 python tools/neural_ab.py > gpurun_out/r3_neural_ab.txt 2>&1; cat gpurun_out/r3_neural_ab.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "neural or golden_losses or random_losses or edge" > gpurun_out/r3_pytest3.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r3_pytest3.log
