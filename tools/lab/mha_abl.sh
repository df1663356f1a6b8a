# usage (GPU box): bash tools/lab/mha_abl.sh tag... -> gpurun_out/mha_abl.txt : forward kernel time per library variant at 256 x 240 (LTRX_MHA_FWD=w32)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/mha_abl.txt
: > $out
for t in "$@"; do
  lib=$R/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$R/allrank_amd/libltrx.so
  rm -rf /tmp/kab
  LTRX_LIB_PATH=$lib LTRX_MHA_FWD=w32 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -o p --output-format csv -- python $R/tools/mha_one.py > /tmp/kab.log 2>&1
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
