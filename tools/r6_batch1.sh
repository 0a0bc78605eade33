#!/bin/bash
# round 6, first GPU call: the new full-size class-table tests, the persistent-loop tests (epoch tags, uncached exchange buffer), the
# loop's time with the buffer in pool / uncached memory and with a cooperative launch, the bootstrap lanes on the persistent loop
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_persist.py -x -q -m gpu 2>&1 | tail -5
for x in uncached pool; do for c in 0 1; do
  echo "=== persist probe: xbuf $x coop $c"
  SFGPU_EM_XBUF=$x SFGPU_EM_COOP=$c EMP_MODES=persist timeout 600 python tools/r5_persist_probe.py 2>&1 | grep -E "==|us/iter" | cut -c1-150
done; done
echo "=== bootstrap lanes (cfg3)"; BSP_SHAPE=cfg3 BSP_N=24 timeout 900 python tools/r6_bs_probe.py 2>&1 | tail -30
echo "=== bootstrap lanes (cfg2)"; BSP_SHAPE=cfg2 BSP_N=24 BSP_XBUF=uncached timeout 600 python tools/r6_bs_probe.py 2>&1 | tail -12
echo "=== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r6_bench_0.json; cut -c1-600 gpurun_out/r6_bench_0.json
} > gpurun_out/r6_batch1.log 2>&1
