#!/bin/bash
# fast inner loop: recompile csrc/w4a16_ring.hip only and relink the product library from the existing objects
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" \
    -c neural_magic_vllm_amd/csrc/w4a16_ring.hip -o build/hip/w4a16_ring.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o neural_magic_vllm_amd/libnmvllm_hip.so build/hip/*.o
echo relinked
