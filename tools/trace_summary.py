"""Summarise a rocprofv3 kernel-trace CSV: device time per RUN of identical launches (same kernel, same grid,
consecutive in time), second half of each run's calls.  Runs are printed in time order, so two benchmark cases
that share kernel and grid (o_proj and down_proj at M = 64) stay on separate lines.
usage: trace_summary.py <kernel_trace.csv> [name regex] [--labels=a,b,c] [--per-config=N]
labels: one per run, in order; --per-config=N: every benchmark case issues exactly N launches (bench_gemm.py: 3 warm
+ 1 pre-capture + 2 graph replays of 40 = 84), so a run of k N launches is k consecutive cases (M = 1 and M = 16
share kernel and grid)"""
import csv
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
labels, per_config = [], 0
for a in sys.argv[1:]:
    if a.startswith("--labels"):
        labels = (a.split("=", 1)[1] if "=" in a else "").split(",")
    if a.startswith("--per-config"):
        per_config = int(a.split("=", 1)[1])
rows = list(csv.DictReader(open(args[0])))
pat = args[1] if len(args) > 1 else "nmv::"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    if not re.search(pat, r["Kernel_Name"]):
        continue
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:78]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(us)
    else:
        runs.append((key, [us]))
if per_config:
    runs = [(k, v[i:i + per_config]) for k, v in runs for i in range(0, len(v), per_config)]
for i, (k, v) in enumerate(runs):
    v = v[len(v) // 2:]
    lab = f"{labels[i]:22s} " if i < len(labels) else ""
    print(f"{lab}{k[0]:78s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):4d} avg={sum(v)/len(v):8.1f}us min={min(v):8.1f}")
