#!/bin/bash
# A/B of the Infinity-Cache weight prefetch (models/llama.py) on the decode step: bench.py at small batches with NMV_PREFETCH=0/1
mkdir -p gpurun_out
for b in ${BATCHES:-1 8 16}; do
  for pf in 0 1; do
    NMV_PREFETCH=$pf NMV_PREFETCH_WGS=${WGS:-64} timeout -k 10 240 python bench.py --batch $b --steps 256 --warmup 8 --no-cpu-baseline --no-sweep \
      > gpurun_out/ab_prefetch_b${b}_pf${pf}.json 2> gpurun_out/ab_prefetch_b${b}_pf${pf}.err || exit 1
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab_prefetch_b${b}_pf${pf}.json") if x.startswith("{")][-1]
d=json.loads(l); print("B=$b prefetch=$pf", d["value"], d["unit"], d["ms_per_step"], "ms/step", flush=True)
PY
  done
done
