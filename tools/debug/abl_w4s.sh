#!/bin/bash
# Ablations of the hand-scheduled 64-row stage of w4a16_stream_kernel (profiles/r03_gemm_ablation.txt).
# Step 1 (build container, CPU only): one development library per switch, the other objects taken from build/hip --
#     bash tools/debug/abl_w4s.sh build
# Step 2 (GPU box; build/abl travels with the snapshot, ~30 MB per library): gate_up at M = 64 with every library --
#     bash tools/debug/abl_w4s.sh
# Results of the ablated builds are garbage; their times are valid.  Remove build/abl afterwards.
VARIANTS="NONE A W LDSRD MFMA EXP PARK BAR"
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "$1" = build ]; then
  mkdir -p "$ROOT/build/abl"
  cd "$ROOT/neural_magic_vllm_amd/csrc" || exit 1
  for v in $VARIANTS AW; do
    defs="-DNMV_W4S_ABL_$v"; [ $v = AW ] && defs="-DNMV_W4S_ABL_A -DNMV_W4S_ABL_W"
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $defs -c w4a16_stream.hip -o "$ROOT/build/abl/w4s_$v.o" 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/abl/lib_$v.so" "$ROOT/build/abl/w4s_$v.o" $(ls "$ROOT"/build/hip/*.o | grep -v w4a16_stream.o) &&
      rm -f "$ROOT/build/abl/w4s_$v.o" && echo "built $v" ) &
  done
  wait
  exit 0
fi
cd "$ROOT" || exit 1
for v in NONE A W AW LDSRD MFMA EXP PARK BAR; do
  echo "== $v"
  NMV_HIP_LIB=build/abl/lib_$v.so python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep gate_up
done
