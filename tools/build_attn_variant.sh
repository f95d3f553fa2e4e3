#!/bin/bash
# A/B builds of the attention kernels: libmla_hip.so with attention.hip compiled under extra flags, all other objects from the product build.
# Usage: bash tools/build_attn_variant.sh <tag> "<extra flags>" [source file]   ->   mla_amd/csrc/build_exp/<tag>/libmla_hip.so  (git-ignored, travels with gpurun)
set -e
TAG=$1; EXTRA=$2; SRC=${3:-attention.hip}
C=$(cd "$(dirname "$0")/../mla_amd/csrc" && pwd)
mkdir -p $C/build_exp/$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $EXTRA -c $C/$SRC -o $C/build_exp/$TAG/attention.o
OBJS=""
for o in $C/build/*.o; do [ "$(basename $o)" = attention.o ] || OBJS="$OBJS $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build_exp/$TAG/libmla_hip.so $OBJS $C/build_exp/$TAG/attention.o
rm -f $C/build_exp/$TAG/attention.o
echo "built $C/build_exp/$TAG/libmla_hip.so"
