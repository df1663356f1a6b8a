# usage (GPU box): bash tools/bench_sweep.sh  -> gpurun_out/sweep.txt  (every bench workload / mode on ONE box, back to back)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/sweep.txt
: > $out
run() {
  python $R/bench.py --no-cpu-baseline --no-side-pass "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %12.1f  ms/step %8.3f  valid/s %s' % ('$*', d['value'], d['ms_per_step'], d.get('valid_items_per_s')))" >> $out
}
run
run --slates-per-gpu 64
run --slates-per-gpu 128
run --dropout 0.1
run --ragged
run --compact
run --compact --dropout 0.1
run --gemm hipblaslt
run --engine autograd
run --workload fc_listnet
run --workload attn_neuralndcg
run --workload attn_lambdarank
run --workload attn1024_listmle
cat $out
