"""Summarise a rocprofv3 kernel-trace CSV: per (kernel, grid) device time, second half of the calls."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "nmv::"
agg = collections.OrderedDict()
for r in rows:
    if not re.search(pat, r["Kernel_Name"]):
        continue
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = v[len(v) // 2:]
    print(f"{k[0]:70s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):4d} avg={sum(v)/len(v):8.1f}us min={min(v):8.1f}")
