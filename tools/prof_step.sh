# usage (on the GPU box): bash tools/prof_step.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>.md
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py --no-cpu-baseline --no-side-pass --steps 20 --warmup 5 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python $R/tools/prof_summary.py $db "bench.py $* (steps 20, warmup 5)" $R/gpurun_out/prof_$tag.md > $R/gpurun_out/prof_$tag.txt
cat $R/gpurun_out/prof_$tag.txt
