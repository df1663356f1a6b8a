# usage (GPU box): bash tools/lab/pmc_ln.sh -> HBM-side counters of the LayerNorm kernels at the bench size (separate pmc passes)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_ln_$i
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_ln_$i -o p --output-format csv -- python $R/tools/ln_one.py > /tmp/pmc_ln_log.txt 2>&1
  f=$(find /tmp/pmc_ln_$i -name "*counter_collection.csv" | head -1)
  k=$(find /tmp/pmc_ln_$i -name "*kernel_trace.csv" | head -1)
  python - "$f" "$k" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "layernorm" in n and "reduce" not in n:
        acc[("fwd" if "fwd" in n else "bwd", r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k[0], k[1], "mean of launches 2..%d: %.5g" % (len(v), sum(v[1:]) / max(len(v) - 1, 1)))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Kernel_Name"]
    if "layernorm" in n and "reduce" not in n:
        dur["fwd" if "fwd" in n else "bwd"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print({k: ["%.1f" % x for x in v] for k, v in dur.items()})
PY
done
