#!/bin/bash
# fast inner loop: recompile one csrc/*.hip file only and relink the product library from the existing objects
# usage: tools/debug/relink_src.sh w4a16_prefill [extra hipcc flags]
set -e
cd "$(dirname "$0")/../.."
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" \
    -c neural_magic_vllm_amd/csrc/$src.hip -o build/hip/$src.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o neural_magic_vllm_amd/libnmvllm_hip.so build/hip/*.o
echo relinked
