# round 5, GPU call 1: new tests (launcher twin, gain function, relu-bits gate, overlap bit-identity, 2-rank overlap) + overlap A/B + trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_main_sequence.py tests/test_gpu_parity.py -x -q -m gpu -k "launcher or gain_function or relu_bits_default or overlap_wgrad or sharded_step or captured" > gpurun_out/r5c1_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r5c1_pytest.log
AB_OPT=overlap_wgrad timeout 600 python tools/wgrad_group_ab.py 64 256 > gpurun_out/r5c1_overlap_ab.log 2>&1; cat gpurun_out/r5c1_overlap_ab.log
cd /tmp && export TMPDIR=/tmp
for ov in 1 0; do
  rm -rf /tmp/tr_$ov
  SLATES=256 STEPS=6 STEP_OPTS=overlap_wgrad=$ov timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_$ov -o p -- python $GRAFT_REPO_ROOT/tools/step_one.py > $GRAFT_REPO_ROOT/gpurun_out/r5c1_trace_$ov.log 2>&1
  db=$(find /tmp/tr_$ov -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/overlap_trace.py $db $GRAFT_REPO_ROOT/gpurun_out/r5c1_overlap_trace_$ov.md; head -8 $GRAFT_REPO_ROOT/gpurun_out/r5c1_overlap_trace_$ov.md
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/r5c1_bench.json 2> gpurun_out/r5c1_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r5c1_bench.json
