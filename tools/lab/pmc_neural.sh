# usage (GPU box): bash tools/lab/pmc_neural.sh -> gpurun_out/pmc_neural.txt : VALU / LDS / wave counters of the NeuralNDCG kernels at the bench shape (separate pmc passes)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_neural.txt
: > $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_neural_$i
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_neural_$i -o p --output-format csv -- python $R/tools/neural_one.py > /tmp/pmc_neural_log.txt 2>&1
  f=$(find /tmp/pmc_neural_$i -name "*counter_collection.csv" | head -1)
  k=$(find /tmp/pmc_neural_$i -name "*kernel_trace.csv" | head -1)
  python - "$f" "$k" >> $out <<'PY'
import csv, sys, collections
def short(n):
    for key in ("forward_blk", "backward_blk", "residual", "pick_iter", "idcg"):
        if key in n:
            return key
    return None
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    s = short(r["Kernel_Name"])
    if s:
        acc[(s, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-13s %-24s %.6g" % (k[0], k[1], sum(v[1:]) / max(len(v) - 1, 1)))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    s = short(r["Kernel_Name"])
    if s:
        dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print("durations us (mean of launches 2..5):", {k: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for k, v in dur.items()})
PY
done
cat $out
