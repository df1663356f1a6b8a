# usage (GPU box): bash tools/lab/mha_ab.sh tag1 tag2 ... -> gpurun_out/mha_ab.txt (attention kernel durations per tools/lab/ab/libltrx_TAG.so, two rounds)
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/mha_ab.txt
for round in 1 2; do for t in "$@"; do
  LTRX_LIB_PATH=$R/tools/lab/ab/libltrx_$t.so bash $R/tools/lab/mha_prof.sh ab_$t
  echo "== $t (round $round)" >> $R/gpurun_out/mha_ab.txt; cat $R/gpurun_out/mha_prof_ab_$t.txt >> $R/gpurun_out/mha_ab.txt
done; done
