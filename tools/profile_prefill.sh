#!/bin/bash
# kernel stats of the 512-token prompt step (TTFT) -> gpurun_out/r04_prefill_step_kernel_stats.csv
set -o pipefail
REPO=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rm -rf /tmp/_pps
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/_pps -- python3 tools/profile_prefill.py ${PROMPT:-512} 6 > gpurun_out/r04_prefill_step.log 2>&1 || { tail -5 gpurun_out/r04_prefill_step.log; exit 1; }
grep "wall ms" gpurun_out/r04_prefill_step.log
f=$(find /tmp/_pps -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("void nmv::", "nmv::", "Cijk", "void at::native"))]
rows = [r for r in rows if "nmv::" in r["Name"] or "Cijk" in r["Name"]]
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
out = open("gpurun_out/r04_prefill_step_kernel_stats.csv", "w")
out.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
for r in rows[:30]:
    name = r["Name"].split("(")[0][:110]
    out.write(f'"{name}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["MinNs"]},{r["MaxNs"]}\n')
    print(f'{name[:90]:90s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  total {int(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY
