cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or mha or varlen or compact or reentrant or encoder" 2>&1 | tail -2
LTRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_dstamp.so timeout 200 python tools/lab/mha_dkdv_stamps.py 2>&1 | grep "^wave"
bash tools/lab/kern_ab.sh "python tools/mha_one.py" main > /dev/null 2>&1; cat gpurun_out/kern_ab.txt
for r in 1 2; do
  echo "== prev"; (cd tools/lab/ab/tree_prev && timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "== new"; timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
