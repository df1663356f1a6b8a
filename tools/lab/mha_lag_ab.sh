# usage (GPU box): bash tools/lab/mha_lag_ab.sh [variant tags ...] -> gpurun_out/mha_lag_ab.txt
# round 6: the phase-shifted second wave of the attention forward (LTRX_MHA_LAG): attention parity tests on the in-tree library, then
# forward / backward kernel durations per library variant (tools/lab/ab/libltrx_TAG.so; "main" = in-tree), LTRX_MHA_FWD=w32, two rounds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/mha_lag_ab.txt
: > $out
cd $R
LTRX_MHA_FWD=w32 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "attention or varlen or mha or model_config3 or dropout" 2>&1 | tail -4 >> $out
cd /tmp
tags="$@"
for round in 1 2; do for t in $tags; do for shape in "256 240" "16 1024" "256 100"; do
  set -- $shape
  lib=$R/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$R/allrank_amd/libltrx.so
  rm -rf /tmp/kab
  LTRX_LIB_PATH=$lib LTRX_MHA_FWD=w32 MB=$1 ML=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -o p --output-format csv -- python $R/tools/mha_one.py > /tmp/kab.log 2>&1
  f=$(find /tmp/kab -name "*kernel_stats.csv" | head -1)
  echo "== $t B=$1 L=$2 (round $round)" >> $out
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx_mha" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done; done
cat $out
