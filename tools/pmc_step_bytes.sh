# usage (GPU box): bash tools/pmc_step_bytes.sh [slates]  -> gpurun_out/pmc_step_bytes_<slates>.md
# HBM-side bytes of ONE training step per kernel family, round-4 switches on vs off: two rocprofv3 --pmc passes per configuration
# (FETCH_SIZE, WRITE_SIZE; counters in KB, FETCH_SIZE x 2 on gfx950 -- MI355X_MICROARCH.md, HBM section), eager steps, step 4 of 4.
S=${1:-64}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in on off; do
  opts=""; [ $cfg = off ] && opts="group_wgrad=0,relu_bits=0,pad_input=0"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/psb_${cfg}_$c
    SLATES=$S STEP_OPTS=$opts timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/psb_${cfg}_$c -o p --output-format csv -- python $R/tools/step_one.py > /dev/null 2>&1
    find /tmp/psb_${cfg}_$c -name "*counter_collection.csv" -exec cp {} /tmp/psb_${cfg}_$c.csv \;
  done
done
python - <<PY > $R/gpurun_out/pmc_step_bytes_$S.md
import csv, collections
def load(cfg):
    out = collections.OrderedDict()
    for c, mul in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
        rows = list(csv.DictReader(open("/tmp/psb_%s_%s.csv" % (cfg, c))))
        # the last step = the launches after the second-to-last adam kernel
        idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
        lo = idx[-2] + 1 if len(idx) >= 2 else 0
        for r in rows[lo:idx[-1] + 1] + rows[idx[-1] + 1:]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
            d = out.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
            d[c] += float(r["Counter_Value"]) * mul
            if c == "FETCH_SIZE": d["n"] += 1
    return out
on, off = load("on"), load("off")
print("# HBM-side bytes of one training step at $S slates x 240 (config 3), per kernel family: round-4 step switches on vs off\n")
print("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (+ --kernel-trace), eager steps, the last of 4; MB = 1e6 bytes; FETCH_SIZE x 2 (gfx950).\n")
print("| kernel | launches on / off | read MB on | read MB off | write MB on | write MB off |\n|---|---|---|---|---|---|")
keys = list(dict.fromkeys(list(on) + list(off)))
tot = [0.0] * 4
for k in sorted(keys, key=lambda k: -(on.get(k, off.get(k))["FETCH_SIZE"] + on.get(k, off.get(k))["WRITE_SIZE"])):
    a, b = on.get(k, {"FETCH_SIZE": 0, "WRITE_SIZE": 0, "n": 0}), off.get(k, {"FETCH_SIZE": 0, "WRITE_SIZE": 0, "n": 0})
    v = [a["FETCH_SIZE"] / 1e6, b["FETCH_SIZE"] / 1e6, a["WRITE_SIZE"] / 1e6, b["WRITE_SIZE"] / 1e6]
    tot = [t + x for t, x in zip(tot, v)]
    if max(v) >= 1.0:
        print("| \`%s\` | %d / %d | %.1f | %.1f | %.1f | %.1f |" % (k, a["n"], b["n"], *v))
print("| **whole step** | | **%.0f** | **%.0f** | **%.0f** | **%.0f** |" % tuple(tot))
PY
cat $R/gpurun_out/pmc_step_bytes_$S.md
