cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layernorm or fused_trainer" 2>&1 | tail -2
bash tools/lab/kern_ab.sh "python tools/ln_one.py" lnold main lng16_512 > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_ln_wide_ab.txt; cat gpurun_out/r3_ln_wide_ab.txt
for r in 1 2; do for t in lnold main lng16_512; do
  lib=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$GRAFT_REPO_ROOT/allrank_amd/libltrx.so
  echo "== $t bench"; LTRX_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
