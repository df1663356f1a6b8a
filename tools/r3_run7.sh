cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "fc_sigmoid_tanh or fused_optimizers or fc_only_relu or falls_back or test_fit_ or fused_options" > gpurun_out/r3_pytest7.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_pytest7.log
bash tools/lab/kern_ab.sh "python tools/ln_one.py" main lng1024 lng2560 > /dev/null 2>&1; grep -v "amdgpu.ids" gpurun_out/kern_ab.txt > gpurun_out/r3_ln_grid_ab.txt; cat gpurun_out/r3_ln_grid_ab.txt
for r in 1 2; do for t in main lng1024 lng2560; do
  lib=tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=allrank_amd/libltrx.so
  LTRX_LIB_PATH=$GRAFT_REPO_ROOT/$lib python bench.py --steps 30 --no-cpu-baseline --no-side-pass 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$t', d['value'], d['ms_per_step'])"
done; done
