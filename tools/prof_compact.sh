# usage (on the GPU box): bash tools/prof_compact.sh <mode>   -> gpurun_out/prof_<mode>.md   (26 steps of tools/compact_timing.py)
mode=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$mode
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o p -- python $R/tools/compact_timing.py $mode > $R/gpurun_out/prof_$mode.log 2>&1
db=$(find /tmp/prof_$mode -name "*.db" | head -1)
python $R/tools/prof_summary.py $db "tools/compact_timing.py $mode (26 steps, 256 ragged slates x 240)" $R/gpurun_out/prof_$mode.md 26 > $R/gpurun_out/prof_$mode.txt
cat $R/gpurun_out/prof_$mode.txt | head -${HEADN:-25}
python $R/tools/prof_by_grid.py $db gemm_nt256 | head -${GRIDN:-40}
if [ -n "$SEQN" ]; then python $R/tools/prof_sequence.py $db $SEQN; fi
