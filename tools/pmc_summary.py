"""Summarise a rocprofv3 counter_collection CSV: mean counter values per (kernel, grid)."""
import collections
import csv
import re
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "nmv::"
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if not re.search(pat, r["Kernel_Name"]):
        continue
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    key = (name, r["Grid_Size"])
    d = agg.setdefault(key, collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    d.setdefault("_dur_us", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(f"{k[0]} grid={k[1]}")
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in d.items()))
