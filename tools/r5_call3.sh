# round 5, GPU call 3: tile-quantisation of the batch -- 240 row tiles of 256 rows on 256 CUs fill 15/16 of every GEMM round
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-side-pass "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$*', '| value', d['value'], 'ms', d['ms_per_step'], 'loss', round(d['last_loss'],5))"; }
for rep in 1 2; do
for b in 256 272 264 273 288 512 544; do run --slates-per-gpu $b; done
done
for b in 64 68 72 128 136; do run --slates-per-gpu $b; done
