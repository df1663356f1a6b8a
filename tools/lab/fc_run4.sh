cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fcstep.py -x -q > gpurun_out/r4_fcstep_tests.log 2>&1; echo tests rc=$?; tail -15 gpurun_out/r4_fcstep_tests.log
timeout 300 python tools/fcstep_check.py --timing-only 64,256,2048 > gpurun_out/fcstep_timing.log 2>&1; cat gpurun_out/fcstep_timing.log
