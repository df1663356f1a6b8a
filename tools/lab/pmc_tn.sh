# usage (GPU box): bash tools/lab/pmc_tn.sh <variant.bin> -> prints L2 / fetch counters of ltrx_gemm_tn256_kernel at GM=61440
v=$1
R=$GRAFT_REPO_ROOT
cp $R/tools/lab/ab/$v.bin $R/allrank_amd/libltrx.so
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_tn_$v_$i
  GM=61440 timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_tn_${v}_$i -o p --output-format csv -- python $R/tools/gemm_one.py > /tmp/pmc_tn_log.txt 2>&1
  f=$(find /tmp/pmc_tn_${v}_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "tn256" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(sys.argv[2], k, "mean over %d launches: %.4g" % (len(v), sum(v[1:]) / max(len(v) - 1, 1)))
PY
  k=$(find /tmp/pmc_tn_${v}_$i -name "*kernel_trace.csv" | head -1)
  python - "$k" "$v" <<'PY'
import csv, sys
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in csv.DictReader(open(sys.argv[1])) if "tn256" in r["Kernel_Name"]]
print(sys.argv[2], "tn256 durations us:", ["%.1f" % x for x in d])
PY
done
