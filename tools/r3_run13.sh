cd $GRAFT_REPO_ROOT
bash tools/lab/kern_ab.sh "python tools/mha_one.py" mhaold main dsoff_early dsoff_dg dsoff_early_dg early > /dev/null 2>&1; cp gpurun_out/kern_ab.txt gpurun_out/r3_mha_ds_kern3.txt; grep -v fwd gpurun_out/r3_mha_ds_kern3.txt
