# usage (GPU box): bash tools/lab/gemm_prec.sh -> gpurun_out/gemm_prec.txt : FFN-shape NT / TN kernel durations per precision code (0 = 3 products, 2 = plain bf16)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/gemm_prec.txt
for prec in 0 2 0 2; do
rm -rf /tmp/gp && GM=61440 GPREC=$prec timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/gp -o p --output-format csv -- python $R/tools/gemm_one.py > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_stats.csv" | head -1)
echo "== prec $prec" >> $R/gpurun_out/gemm_prec.txt
python - "$f" >> $R/gpurun_out/gemm_prec.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ltrx" in r["Name"]:
        print("%-50s calls %s avg_us %.1f min_us %.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
