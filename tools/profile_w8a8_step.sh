#!/bin/bash
# kernel stats of the W8A8 + fp8-KV decode step at B = 64 (BASELINE.json configs[3]) -> gpurun_out/r04_w8a8_bench_kernel_stats.csv
set -o pipefail
REPO=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rm -rf /tmp/_pw8
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_pw8 -- python3 bench.py --quant w8a8 --kv-cache-dtype fp8 --steps 16 --warmup 4 --no-sweep --no-cpu-baseline > gpurun_out/r04_w8a8_step.log 2>&1 || { tail -5 gpurun_out/r04_w8a8_step.log; exit 1; }
f=$(find /tmp/_pw8 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nmv::" in r["Name"] or "Cijk" in r["Name"]]
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
out = open("gpurun_out/r04_w8a8_bench_kernel_stats.csv", "w")
out.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
for r in rows[:24]:
    name = r["Name"].split("(")[0][:110]
    out.write(f'"{name}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["MinNs"]},{r["MaxNs"]}\n')
    print(f'{name[:96]:96s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
