#!/bin/bash
# round 6: the whole GPU suite + the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "=== pytest -m gpu"; timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
echo "=== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r6_bench_1.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_bench_1.json"))
for k in ("value", "ms_per_step", "phase_ms", "em_iters", "em_us_per_iter_loop", "roofline", "roofline_em_iteration", "roofline_class_build", "bootstrap_ms_per_replicate", "gibbs_1000_draws_s", "parity_vs_cpu"):
    if k in d: print(k, ":", json.dumps(d[k])[:400])
PY
} > gpurun_out/r6_full.log 2>&1
