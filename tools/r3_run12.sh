cd $GRAFT_REPO_ROOT
bash tools/lab/kern_ab.sh "python tools/mha_one.py" main dsoff dshalf > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_mha_ds_kern2.txt; grep -v fwd gpurun_out/r3_mha_ds_kern2.txt
