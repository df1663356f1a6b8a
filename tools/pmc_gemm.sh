# usage (GPU box): bash tools/pmc_gemm.sh  -> gpurun_out/pmc_gemm/*.csv : FETCH_SIZE | WRITE_SIZE | SQ counters of tools/gemm_one.py (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_gemm
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_gemm/p$i -o p$i --output-format csv -- python $R/tools/gemm_one.py > $R/gpurun_out/pmc_gemm/log$i.txt 2>&1
  find /tmp/pmc_gemm/p$i -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/pmc_gemm/p$i.csv \;
  find /tmp/pmc_gemm/p$i -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/pmc_gemm/k$i.csv \;
done
ls $R/gpurun_out/pmc_gemm
