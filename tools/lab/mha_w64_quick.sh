# usage (GPU box): bash tools/lab/mha_w64_quick.sh -> gpurun_out/mha_w64_quick.txt : attention tests + stamps + forward kernel time (w64, w32)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/mha_w64_quick.txt
: > $out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "attention or varlen or mha or model_config3 or dropout" 2>&1 | tail -4 >> $out
cd /tmp
echo "== stamps" >> $out
LTRX_LIB_PATH=$R/tools/lab/ab/libltrx_stamp.so timeout 300 python $R/tools/lab/mha_w64_stamps.py 2>&1 | grep -v amdgpu.ids | head -24 >> $out
for m in w64 w32; do for shape in "256 240" "16 1024" "256 100"; do
  set -- $shape
  rm -rf /tmp/kab
  LTRX_MHA_FWD=$m MB=$1 ML=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -o p --output-format csv -- python $R/tools/mha_one.py > /tmp/kab.log 2>&1
  f=$(find /tmp/kab -name "*kernel_stats.csv" | head -1)
  echo "== $m B=$1 L=$2" >> $out
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx_mha_fwd" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done
cat $out
