#!/bin/bash
# Ablation builds of csrc/w4a16_ring.hip (NMV_W4R_ABL_* switches: results garbage, times valid): one .so per variant under
# build/abl/, linked from the product build's other objects; run with NMV_HIP_LIB=build/abl/libnmv_<variant>.so.
# usage: tools/debug/abl_w4r.sh [variant=flags ...]   e.g.  nosums=-DNMV_W4R_ABL_SUMS
set -e
cd "$(dirname "$0")/../.."
OBJS=$(ls build/hip/*.o | grep -v w4a16_ring.o)
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result ${flags//,/ } \
      -c neural_magic_vllm_amd/csrc/w4a16_ring.hip -o build/abl/ring_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libnmv_$name.so $OBJS build/abl/ring_$name.o
  echo "built build/abl/libnmv_$name.so ($flags)"
done
