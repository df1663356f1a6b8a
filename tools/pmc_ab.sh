# usage (GPU box): bash tools/pmc_ab.sh <tag> <STEP_OPTS on> <STEP_OPTS off> -> gpurun_out/pmc_ab_<tag>.md
# per-(kernel, grid) means of SQ_INSTS_VALU / SQ_INSTS_MFMA / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES over eager training steps of
# tools/step_one.py (SLATES=256) with a FusedTrainer switch on and off: one rocprofv3 --pmc pass each (counters only beside --kernel-trace)
tag=$1; on=$2; off=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in on off; do
  opts=$on; [ $v = off ] && opts=$off
  rm -rf /tmp/pmcab_${tag}_$v
  SLATES=${SLATES:-256} STEPS=3 STEP_OPTS=$opts timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmcab_${tag}_$v -o p --output-format csv -- python $R/tools/step_one.py > $R/gpurun_out/pmc_ab_${tag}_$v.log 2>&1
  find /tmp/pmcab_${tag}_$v -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/pmc_ab_${tag}_$v.csv \;
done
python - <<PY
import csv, collections
def load(f):
    t = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:58], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        t[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return t
on, off = load("$R/gpurun_out/pmc_ab_${tag}_on.csv"), load("$R/gpurun_out/pmc_ab_${tag}_off.csv")
m = lambda d, c: (sum(d[c]) / len(d[c])) if d.get(c) else float("nan")
out = open("$R/gpurun_out/pmc_ab_${tag}.md", "w")
out.write("# PMC A/B ($tag): STEP_OPTS=$on vs $off, tools/step_one.py SLATES=${SLATES:-256}, means per launch\n\n")
out.write("| kernel | grid | launches | VALU insts on | off | on/off | MFMA insts on | off | busy cycles on | off | on/off |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
for k in sorted(on, key=lambda k: -m(on[k], "SQ_BUSY_CYCLES") * len(on[k].get("SQ_BUSY_CYCLES", []))):
    if k not in off or not on[k].get("SQ_INSTS_VALU"):
        continue
    a, b = on[k], off[k]
    line = "| %s | %s | %d | %.4g | %.4g | %.3f | %.4g | %.4g | %.4g | %.4g | %.3f |" % (
        "\`" + k[0] + "\`", k[1], len(a["SQ_INSTS_VALU"]), m(a, "SQ_INSTS_VALU"), m(b, "SQ_INSTS_VALU"), m(a, "SQ_INSTS_VALU") / max(m(b, "SQ_INSTS_VALU"), 1),
        m(a, "SQ_INSTS_MFMA"), m(b, "SQ_INSTS_MFMA"), m(a, "SQ_BUSY_CYCLES"), m(b, "SQ_BUSY_CYCLES"), m(a, "SQ_BUSY_CYCLES") / max(m(b, "SQ_BUSY_CYCLES"), 1))
    print(line); out.write(line + "\n")
PY
