cd /root/repo
run() { python bench.py --steps 30 --no-cpu-baseline --no-side-pass "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$*', '| value', d['value'], 'ms', d['ms_per_step'], 'valid', d.get('valid_items_per_s'), 'loss', round(d['last_loss'],5))"; }
run
run --slates-per-gpu 512
run --slates-per-gpu 64
run --slates-per-gpu 128
run --dropout 0.1
run --ragged
run --ragged --compact
run --ragged --compact --dropout 0.1
run --workload attn_neuralndcg
run --workload attn_lambdarank
run --workload attn1024_listmle
run --workload fc_listnet
run --gemm hipblaslt
run --gemm split_bf16_strict
run --gemm bf16
run --gemm bf16 --slates-per-gpu 512
run --gemm bf16 --ragged --compact
