#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default decode bench at B = 64 and B = 1 -> gpurun_out/r04_bench_kernel_stats[_b1].csv (nmv kernels only)
set -o pipefail
REPO=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
for b in 64 1; do
  rm -rf /tmp/_pbs
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_pbs -- python3 bench.py --steps 16 --warmup 4 --no-sweep --no-cpu-baseline --batch $b > gpurun_out/r04_bench_stats_b$b.log 2>&1 || { tail -5 gpurun_out/r04_bench_stats_b$b.log; exit 1; }
  f=$(find /tmp/_pbs -name "*kernel_stats.csv" | head -1)
  suffix=$([ $b = 64 ] && echo "" || echo "_b1")
  cp $f gpurun_out/r04_bench_kernel_stats$suffix.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nmv::" in r["Name"]]
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f'{r["Name"].split("(")[0][:90]:90s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:7.2f} us  min {int(r["MinNs"]) / 1e3:7.2f}  max {int(r["MaxNs"]) / 1e3:7.2f}')
PY
done
