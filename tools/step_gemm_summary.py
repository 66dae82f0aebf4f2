"""Per-projection summary of tools/bench_step_gemms.py under rocprofv3: kernel-trace CSV -> mean device time per launch;
PMC CSVs (FETCH_SIZE, WRITE_SIZE: one counter per pass) -> HBM-side bytes per launch, FETCH doubled (gfx950 tallies the
128-byte requests of wide streaming reads at 64 B: MI355X_MICROARCH.md, HBM), both counters in KiB.
usage: python tools/step_gemm_summary.py --ms 32,64 --groups 24 [--trace t.csv] [--fetch f.csv --write w.csv] [--json out.json]"""
import argparse
import collections
import csv
import json
import re

SHAPES = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]


def gemm_rows(path, value_of):
    rows = []
    for r in csv.DictReader(open(path)):
        if not re.search(r"w4a16_(ring|stream|gemm)", r["Kernel_Name"]):
            continue
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], value_of(r)))
    rows.sort()
    return rows


def split(rows, ms, groups):
    """bench_step_gemms.py order: for m: for group: qkv, o, gate_up, down"""
    assert len(rows) == len(ms) * groups * 4, (len(rows), len(ms), groups)
    out = collections.OrderedDict()
    it = iter(rows)
    for m in ms:
        acc = {name: [] for name, _, _ in SHAPES}
        kern = {}
        for _ in range(groups):
            for name, _, _ in SHAPES:
                _, kn, v = next(it)
                acc[name].append(v)
                kern[name] = re.sub(r"\(.*", "", kn.replace("void ", "")).replace("nmv::", "")
        out[m] = (acc, kern)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="32,64")
    ap.add_argument("--groups", type=int, default=24)
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--json")
    a = ap.parse_args()
    ms = [int(x) for x in a.ms.split(",")]
    res = collections.OrderedDict((str(m), collections.OrderedDict()) for m in ms)
    if a.trace:
        t = split(gemm_rows(a.trace, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), ms, a.groups)
        for m, (acc, kern) in t.items():
            for name, _, _ in SHAPES:
                v = acc[name][2:]   # the first launches touch cold code / tickets
                res[str(m)].setdefault(name, {}).update(kernel=kern[name], us=round(sum(v) / len(v), 2), us_min=round(min(v), 2))
    if a.fetch and a.write:
        f = split(gemm_rows(a.fetch, lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "FETCH_SIZE" else None), ms, a.groups)
        w = split(gemm_rows(a.write, lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "WRITE_SIZE" else None), ms, a.groups)
        for m in ms:
            for name, k, n in SHAPES:
                fb = 2 * 1024 * sum(f[m][0][name]) / a.groups
                wb = 1024 * sum(w[m][0][name]) / a.groups
                alg = k * n // 2 + (k // 128) * n * 2 + 2 * m * k + 2 * m * (n // 2 if name == "gate_up" else n)
                res[str(m)].setdefault(name, {}).update(fetch_bytes=int(fb), write_bytes=int(wb), algorithmic_bytes=alg,
                                                        ratio=round((fb + wb) / alg, 3))
    for m, d in res.items():
        tot_us = sum(v.get("us", 0) for v in d.values())
        tot_b = sum(v.get("fetch_bytes", 0) + v.get("write_bytes", 0) for v in d.values())
        tot_alg = sum(v.get("algorithmic_bytes", 0) for v in d.values())
        print(f"M={m}: " + "  ".join(f"{k}: " + " ".join(f"{kk}={vv}" for kk, vv in v.items() if kk != "kernel") for k, v in d.items()))
        if tot_us:
            print(f"   group {tot_us:.1f} us" + (f", {tot_b} bytes = {tot_b / tot_alg:.3f} x algorithmic ({tot_alg})" if tot_alg else ""))
        d["_group"] = {"us": round(tot_us, 2), "bytes": tot_b, "algorithmic_bytes": tot_alg,
                       "ratio_to_algorithmic": round(tot_b / tot_alg, 3) if tot_alg else None}
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
