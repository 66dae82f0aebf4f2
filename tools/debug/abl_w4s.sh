set -e
for v in NONE A W AW LDSRD MFMA EXP PARK BAR; do
  echo "== $v"
  NMV_HIP_LIB=build/abl/lib_$v.so python tools/bench_gemm.py --native --ms 64 --shapes gate_up 2>&1 | grep gate_up
done
