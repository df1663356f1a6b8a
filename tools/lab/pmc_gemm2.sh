# usage (GPU box): bash tools/lab/pmc_gemm2.sh -> gpurun_out/pmc_gemm2.txt : matrix-pipe / LDS / HBM counters of the large-tile GEMM kernels at the FFN shape (separate pmc passes)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_gemm2.txt
: > $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_gemm2_$i
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_gemm2_$i -o p --output-format csv -- env GM=61440 python $R/tools/gemm_one.py > /tmp/pmc_gemm2_log.txt 2>&1
  f=$(find /tmp/pmc_gemm2_$i -name "*counter_collection.csv" | head -1)
  k=$(find /tmp/pmc_gemm2_$i -name "*kernel_trace.csv" | head -1)
  python - "$f" "$k" >> $out <<'PY'
import csv, sys, collections
def short(n):
    return "nt256" if "nt256" in n else ("tn256" if "tn256" in n else None)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    s = short(r["Kernel_Name"])
    if s:
        acc[(s, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-5s %-28s %.6g" % (k[0], k[1], sum(v[1:]) / max(len(v) - 1, 1)))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    s = short(r["Kernel_Name"])
    if s:
        dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print("durations us (mean of launches 2..5):", {k: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for k, v in dur.items()})
PY
done
cat $out
