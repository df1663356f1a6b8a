# usage (GPU box): bash tools/lab/ab_prev.sh [pytest -k expr] -- the working tree against tools/lab/ab/tree_prev (an export of the previous commit,
# built): selected GPU tests of the working tree, kernel stats of the default bench for both, three interleaved bench rounds
cd $GRAFT_REPO_ROOT
[ -n "$1" ] && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" 2>&1 | tail -2
bash tools/prof_step.sh ab_new > /dev/null 2>&1; echo "== new"; grep "${2:-gemm_nt256}" gpurun_out/prof_ab_new.md | grep "^| \`void ltrx_gemm\|983040\|245760\|737280" | cut -c1-110 | head -8
(cd tools/lab/ab/tree_prev && GRAFT_REPO_ROOT=$PWD bash tools/prof_step.sh ab_prev > /dev/null 2>&1; echo "== prev"; grep "${2:-gemm_nt256}" gpurun_out/prof_ab_prev.md | grep "^| \`void ltrx_gemm\|983040\|245760\|737280" | cut -c1-110 | head -8)
for r in 1 2 3; do
  echo "== prev"; (cd tools/lab/ab/tree_prev && timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "== new"; timeout 300 python bench.py --no-cpu-baseline --no-side-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
