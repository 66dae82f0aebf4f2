#!/bin/bash
# smoke + the default bench line, summarised
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 300 gpurun_out/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_final.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["weights_gb_per_rank"])
print({k: d[k] for k in ("ttft_ms_p50", "batch_sweep_tokens_per_s", "other_configs") if k in d})
r = d["roofline"]
print(r["frac"], r["avg_us_per_launch_group"], r["per_gemm_us"], r["reference_op"]["frac"])
PY
