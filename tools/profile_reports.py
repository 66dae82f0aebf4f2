"""Turn the raw rocprofv3 CSVs that `tools/gpu_ci.sh prof attnprof attntraffic` leave under gpurun_out/ into the
committed summaries profiles/rNN_bench_kernel_stats.csv and profiles/rNN_attention.txt.
usage: python tools/profile_reports.py r02"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def bench_stats():
    rows = list(csv.DictReader(open(os.path.join(OUT, "kernel_stats.csv"))))
    keep = [r for r in rows if "nmv::" in r["Name"] or r["Name"].startswith("Cijk") or "rocclr" in r["Name"]]
    with open(os.path.join(PROF, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 16 --warmup 4 --no-sweep --no-cpu-baseline  (MI355X)\n"
                "# rows of the decode step and its set-up launches only (nmv:: kernels, the hipBLASLt lm_head GEMM, runtime copies); the\n"
                "# at::native kernels that build the synthetic weights and KV context before the timed region are left out.  22 steps x 32 layers:\n"
                "# 704 attention launches, 2904 = 3 x 968 qkv/o/down GEMMs (deferred reduction), 968 gate_up GEMMs (silu-mul epilogue), 1408 norms.\n")
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            w.writerow(r)
    for r in keep[:6]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg={float(r['AverageNs']) / 1e3:8.1f} us")


def rows_of(path, pat, counter=None):
    rs = list(csv.DictReader(open(path)))
    if counter:
        rs = [r for r in rs if r["Counter_Name"] == counter]
    rs = [r for r in rs if re.search(pat, r["Kernel_Name"])]
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rs


def attention(cases=((512, "plain"), (512, "fused prologue (2 qkv slabs)"), (530, "plain"),
                     (530, "fused prologue (2 qkv slabs)"), (650, "plain"), (650, "fused prologue (2 qkv slabs)"))):
    tr = rows_of(os.path.join(OUT, "attn_trace.csv"), "paged_attention_kernel")
    ft = rows_of(os.path.join(OUT, "attn_fetch.csv"), "paged_attention_kernel", "FETCH_SIZE")
    out = ["# rocprofv3 --kernel-trace / --pmc FETCH_SIZE (separate runs) -- python tools/bench_attn.py --cases 64:512,64:530,64:650 --fused 2",
           "# MI355X.  paged_attention_kernel<BF16, kv auto, D=128, block 16, 4 heads per group, 4 waves>, grid (8 groups, 64 seqs): Llama-3-8B",
           "# geometry, B = 64, random block tables over a 640 MB cache.  42 launches per case (1 warm + 1 pre-capture + 2 graph replays of 20),",
           "# second half averaged.  algorithmic bytes = 2 * L * 8 kv heads * 128 * 2 B * 64 seqs; FETCH_SIZE in KiB x 2 (gfx950 wide-read correction).",
           f"{'case':44s} {'avg us':>8s} {'min us':>8s} {'alg MB':>8s} {'alg TB/s':>9s} {'fetch MB':>9s} {'fetch/alg':>9s}"]
    for i, (L, form) in enumerate(cases):
        t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr[i * 42:(i + 1) * 42]][21:]
        fv = [float(r["Counter_Value"]) for r in ft[i * 42:(i + 1) * 42]][21:]
        alg = 2 * L * 8 * 128 * 2 * 64
        fb = 2 * 1024 * sum(fv) / len(fv)
        avg = sum(t) / len(t)
        out.append(f"{'B=64 L=%d %s' % (L, form):44s} {avg:8.1f} {min(t):8.1f} {alg / 1e6:8.1f} {alg / avg / 1e6:9.2f} "
                   f"{fb / 1e6:9.1f} {fb / alg:9.2f}")
    open(os.path.join(PROF, f"{tag}_attention.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[4:]))


if __name__ == "__main__":
    bench_stats()
    attention()
