#!/bin/bash
# scratch driver of one gpurun call (edited per call; see tools/gpu_ci.sh for the standing steps)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for b in 64 1; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$b -- python bench.py --steps 16 --warmup 4 --no-sweep --no-cpu-baseline --batch $b > gpurun_out/prof_b$b.log 2>&1
  find gpurun_out/prof_b$b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_b$b.csv
  rm -rf gpurun_out/prof_b$b
done
python -c "import __graft_entry__ as g; g.smoke()"
