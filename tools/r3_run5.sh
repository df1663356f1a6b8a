cd $GRAFT_REPO_ROOT
for round in 1 2; do for t in main fence rcp wide rcpwide; do
  lib=tools/lab/ab/libltrx_$t.so; [ "$t" = main ] && lib=allrank_amd/libltrx.so
  echo "== $t (round $round)"; LTRX_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout 200 python tools/neural_ab.py 2>&1 | grep "^{"
done; done > gpurun_out/r3_neural_variants.txt 2>&1
cat gpurun_out/r3_neural_variants.txt
