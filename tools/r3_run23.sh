cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  echo "== prev"; (cd tools/lab/ab/tree_prev && timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "== new"; timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
