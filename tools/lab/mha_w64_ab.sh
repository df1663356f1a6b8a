# usage (GPU box): bash tools/lab/mha_w64_ab.sh -> gpurun_out/mha_w64_ab.txt
# round 6: the four-wave x 64-query attention forward (default) against the eight-wave x 32-query kernel (LTRX_MHA_FWD=w32): parity
# tests of both, then kernel durations (rocprofv3 --kernel-trace --stats of tools/mha_one.py, config-3 and config-5 shapes), two rounds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/mha_w64_ab.txt
: > $out
cd $R
for m in w64 w32; do
  echo "== pytest attention ($m)" >> $out
  LTRX_MHA_FWD=$m timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "attention or varlen or mha or model_config3 or dropout" 2>&1 | tail -4 >> $out
done
cd /tmp
for round in 1 2; do for m in w64 w32; do for shape in "256 240" "16 1024" "256 100"; do
  set -- $shape
  rm -rf /tmp/kab
  LTRX_MHA_FWD=$m MB=$1 ML=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -o p --output-format csv -- python $R/tools/mha_one.py > /tmp/kab.log 2>&1
  f=$(find /tmp/kab -name "*kernel_stats.csv" | head -1)
  echo "== $m B=$1 L=$2 (round $round)" >> $out
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx_mha" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done; done
cat $out
