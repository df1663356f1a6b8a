# usage (here): bash tools/lab/lib_variant.sh SRC TAG "-DFLAG=1 ..."  -> tools/lab/ab/libltrx_TAG.so = libltrx.so with allrank_amd/csrc/SRC.hip
# recompiled under the extra flags (the other objects come from allrank_amd/build/); run with LTRX_LIB_PATH=...
set -e
R=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $R/tools/lab/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $3 -Rpass-analysis=kernel-resource-usage -c $R/allrank_amd/csrc/$1.hip -o /tmp/$1_$2.o 2>&1 | grep "Function Name\| VGPRs:\|VGPRs Spill" | paste - - - | sed 's/.*Function Name: \([^ ]*\).*VGPRs: \([0-9]*\).*Spill: \([0-9]*\).*/\1 vgpr \2 spill \3/' | grep "${4:-.}" | cut -c1-100
objs=$(ls $R/allrank_amd/build/*.o | grep -v "/$1.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lab/ab/libltrx_$2.so $objs /tmp/$1_$2.o
echo "built $2"
