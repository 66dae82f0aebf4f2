#!/bin/bash
# Ablation builds of one csrc/*.hip file (its -D development switches: results garbage, times valid): one .so per variant
# under build/abl/, linked from the product build's other objects; run with NMV_HIP_LIB=build/abl/libnmv_<variant>.so.
# usage: tools/debug/abl_src.sh w4a16_prefill name=-DNMV_W4P_ABL=1 [name=flags ...]
set -e
cd "$(dirname "$0")/../.."
src=$1; shift
mkdir -p build/abl
OBJS=$(ls build/hip/*.o | grep -v "/$src.o")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result ${flags//,/ } \
      -c neural_magic_vllm_amd/csrc/$src.hip -o build/abl/${src}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libnmv_$name.so $OBJS build/abl/${src}_$name.o
  echo "built build/abl/libnmv_$name.so ($flags)"
done
