"""Per-kernel register / LDS / spill table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py neural_magic_vllm_amd/csrc/w4a16_stream.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(nmv::GemmParams\)|nmv::|void ", "", name)
    print(f"{name[:90]:90s} vgpr {r.get('VGPRs','?'):>3s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs', r.get('SGPRs','?')):>3s} "
          f"spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')} "
          f"lds {r.get('LDS Size [bytes/block]','?')}")
