"""print a window of a rocprofv3 --kernel-trace CSV as a timeline (start, duration, queue, name): who overlaps whom.
usage: trace_window.py trace.csv <row | name-substring[:occurrence]> <count>"""
import csv, sys
path, anchor, count = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if anchor.lstrip("-").isdigit():
    skip = int(anchor)
    skip = skip if skip >= 0 else len(rows) + skip
else:
    pat, _, occ = anchor.partition(":")
    hits = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
    print(f"{len(hits)} launches of *{pat}*")
    skip = hits[int(occ) if occ else len(hits) // 2]
t0 = int(rows[skip]["Start_Timestamp"])
for r in rows[skip:skip + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.2f} us  +{(e - s) / 1e3:7.2f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
