# usage (on the GPU box): bash tools/pmc_step.sh <tag>  -> gpurun_out/pmc_<tag>.txt  (per-kernel means of a few SQ counters)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$tag/p$i -o p$i --output-format csv -- python $R/bench.py --no-cpu-baseline --no-side-pass --steps 2 --warmup 3 "$@" > $R/gpurun_out/pmc_$tag/log$i.txt 2>&1
  cp /tmp/pmc_$tag/p$i/*counter_collection.csv $R/gpurun_out/pmc_$tag/p$i.csv 2>/dev/null || find /tmp/pmc_$tag/p$i -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/pmc_$tag/p$i.csv \;
done
python - <<PY
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/pmc_$tag/p*.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out=open("$R/gpurun_out/pmc_$tag.txt","w")
for k,d in sorted(tot.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES",[0]))):
    if not d.get("SQ_BUSY_CYCLES"): continue
    line="%-62s n=%d "%(k,len(d["SQ_BUSY_CYCLES"]))+" ".join("%s=%.3g"%(c.replace("SQ_",""),sum(v)/len(v)) for c,v in sorted(d.items()))
    print(line); out.write(line+"\n")
PY
