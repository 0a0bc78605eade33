#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multinomial or bootstrap" 2>&1 | tail -3
for shape in cfg3 cfg2; do
  echo "=== $shape, fused tree";  BSP_SHAPE=$shape BSP_N=24 BSP_XBUF=uncached BSP_LANESETS=1:1,3:0 timeout 600 python tools/r6_bs_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
  echo "=== $shape, tree by levels"; SFGPU_MN_TREE=levels BSP_SHAPE=$shape BSP_N=24 BSP_XBUF=uncached BSP_LANESETS=1:1,3:0 timeout 600 python tools/r6_bs_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
done
} > gpurun_out/r6_bs2.log 2>&1
