cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/fcstep_check.py --quick > gpurun_out/fcstep_check.log 2>&1; echo check rc=$?
for v in fcbtop; do
  echo "== $v" >> gpurun_out/fcstep_check.log
  LTRX_LIB_PATH=tools/lab/ab/libltrx_$v.so timeout 120 python tools/fcstep_check.py --timing-only 256,2048 2>&1 | grep fcstep >> gpurun_out/fcstep_check.log
done
LTRX_LIB_PATH=tools/lab/ab/libltrx_fcstamp.so timeout 120 python tools/lab/fc_stamps.py 2048 > gpurun_out/fc_stamps_2048.txt 2>&1
cat gpurun_out/fcstep_check.log; grep -A9 "wave 0" gpurun_out/fc_stamps_2048.txt | head -10; grep -A9 "wave 11" gpurun_out/fc_stamps_2048.txt
