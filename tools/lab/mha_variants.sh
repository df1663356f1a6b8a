# usage (here): bash tools/lab/mha_variants.sh TAG "-DLTRX_MHA_...=1 ..."  -> tools/lab/ab/libltrx_TAG.so (the other objects come from allrank_amd/build/)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/tools/lab/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $2 -Rpass-analysis=kernel-resource-usage -c $R/allrank_amd/csrc/ltrx_mha_res.hip -o /tmp/mha_res_$1.o 2>&1 | grep "VGPRs Spill\|error" | tr '\n' ' '
objs=$(ls $R/allrank_amd/build/*.o | grep -v ltrx_mha_res.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lab/ab/libltrx_$1.so $objs /tmp/mha_res_$1.o
echo " built $1"
