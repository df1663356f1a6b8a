# usage (GPU box): bash tools/lab/mha_prof.sh TAG -> gpurun_out/mha_prof_TAG.txt : average duration of the attention kernels (mha_one.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mp -o p --output-format csv -- python $R/tools/mha_one.py > /tmp/mp.log 2>&1
f=$(find /tmp/mp -name "*kernel_stats.csv" | head -1)
python - "$f" > $R/gpurun_out/mha_prof_$1.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx" in r["Name"]:
        print("%-40s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
if [ "$2" = "test" ]; then cd $R && python -m pytest tests/test_gpu_parity.py -q -x -k "attention" 2>&1 | tail -3; fi
