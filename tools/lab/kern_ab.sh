# usage (GPU box): bash tools/lab/kern_ab.sh "ENV=.. python-tool.py" tag1 tag2 ... -> gpurun_out/kern_ab.txt
# kernel durations (rocprofv3 --kernel-trace --stats) of one tool run per library variant tools/lab/ab/libltrx_TAG.so
# ("main" = the in-tree allrank_amd/libltrx.so), two rounds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cmd=$1; shift
: > $R/gpurun_out/kern_ab.txt
for round in 1 2; do for t in "$@"; do
  lib=$R/tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=$R/allrank_amd/libltrx.so
  rm -rf /tmp/kab
  env LTRX_LIB_PATH=$lib $(echo $cmd | sed "s#python #timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -o p --output-format csv -- python $R/#") > /tmp/kab.log 2>&1
  f=$(find /tmp/kab -name "*kernel_stats.csv" | head -1)
  echo "== $t (round $round)" >> $R/gpurun_out/kern_ab.txt
  python - "$f" >> $R/gpurun_out/kern_ab.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done; done
cat $R/gpurun_out/kern_ab.txt
