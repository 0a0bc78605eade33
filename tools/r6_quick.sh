#!/bin/bash
# round 6: the EM / builder tests that matter + the default bench line's phases
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "=== tests"; timeout 1500 python -m pytest tests/test_gpu_persist.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5
echo "=== bench"; timeout 900 python bench.py ${BENCH_ARGS:-} 2>&1 | tail -1 > gpurun_out/r6_bench_q.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_bench_q.json"))
for k in ("value", "ms_per_step", "phase_ms", "em_iters", "em_us_per_iter_loop", "bootstrap_ms_per_replicate", "gibbs_1000_draws_s"):
    if k in d: print(k, ":", json.dumps(d[k])[:400])
print("em overhead ms:", d["phase_ms"]["em"] - d["em_iters"] * d["em_us_per_iter_loop"] * 1e-3, " class build - kernels:", d["phase_ms"]["class_build"] - d["phase_ms"]["insert_kernel"])
PY
} > gpurun_out/r6_quick.log 2>&1
