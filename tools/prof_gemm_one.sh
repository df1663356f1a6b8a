# usage (GPU box): GM=61440 bash tools/prof_gemm_one.sh -> gpurun_out/prof_gemm_one.md (kernel-trace summary of the FFN-shape GEMMs alone)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_g1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g1 -o p -- python $R/tools/gemm_one.py > $R/gpurun_out/prof_gemm_one.log 2>&1
db=$(find /tmp/prof_g1 -name "*.db" | head -1)
python $R/tools/prof_summary.py $db "tools/gemm_one.py GM=$GM (NT: [GM,512]x[2048,512]^T + bias; TN: dW[2048,512] = dY^T X), 5 launches each" $R/gpurun_out/prof_gemm_one.md
