# usage (GPU box): bash tools/lab/mha_w64_probe.sh -> gpurun_out/mha_w64_probe.txt
# round 6: cycle stamps of the 64-query forward + its ablations (no MFMAs / no softmax arithmetic in the pipelined iteration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/mha_w64_probe.txt
: > $out
echo "== stamps" >> $out
LTRX_LIB_PATH=$R/tools/lab/ab/libltrx_stamp.so timeout 300 python $R/tools/lab/mha_w64_stamps.py >> $out 2>&1
for t in main nomfma novalu; do
  lib=$R/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$R/allrank_amd/libltrx.so
  rm -rf /tmp/kab
  LTRX_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -o p --output-format csv -- python $R/tools/mha_one.py > /tmp/kab.log 2>&1
  f=$(find /tmp/kab -name "*kernel_stats.csv" | head -1)
  echo "== $t" >> $out
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx_mha_fwd" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
cat $out
