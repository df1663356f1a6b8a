# usage (GPU box): bash tools/lab/step_ab.sh tagA tagB [bench args...] -> gpurun_out/step_ab.txt
# the training step (bench.py --steps 60, no side passes) with two library variants (tools/lab/ab/libltrx_TAG.so; "main" = in-tree), interleaved, two rounds
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift; shift
out=gpurun_out/step_ab.txt
: > $out
for round in 1 2; do for t in $A $B; do
  lib=$GRAFT_REPO_ROOT/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$GRAFT_REPO_ROOT/allrank_amd/libltrx.so
  for extra in "" "--dropout 0.1" "--workload attn1024_listmle" "--ragged --compact"; do
    LTRX_LIB_PATH=$lib python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-pass "$@" $extra 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$t round $round [$extra]', 'value', d['value'], 'ms', d['ms_per_step'], 'loss', round(d['last_loss'],6))" >> $out
  done
done; done
cat $out
