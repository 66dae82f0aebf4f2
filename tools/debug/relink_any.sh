#!/bin/bash
# recompile the given csrc/*.hip files and relink the product library: tools/debug/relink_any.sh scaled_mm w4a16_prefill
set -e
cd "$(dirname "$0")/../.."
for src in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c neural_magic_vllm_amd/csrc/$src.hip -o build/hip/$src.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o neural_magic_vllm_amd/libnmvllm_hip.so build/hip/*.o
echo relinked
