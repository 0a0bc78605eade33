#!/bin/bash
# round 6: why do bootstrap lanes on the persistent loop give up?  (-DSFGPU_P_PROGRESS build: per-tile progress, wait reasons, start times)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_progress.so BSP_SHAPE=cfg2 BSP_N=6 BSP_XBUF=uncached timeout 600 python tools/r6_bs_probe.py > gpurun_out/r6_bs_dbg.log 2>&1
