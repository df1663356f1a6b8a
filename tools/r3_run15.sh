cd $GRAFT_REPO_ROOT
for r in 1 2; do
for t in mhaold main; do
  lib=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$GRAFT_REPO_ROOT/allrank_amd/libltrx.so
  echo "== $t bench"; LTRX_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
LTRX_LIB_PATH=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_mhaold.so bash tools/prof_step.sh r3_mhaold | head -16
bash tools/prof_step.sh r3_mhanew | head -16
