#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{ EMP_SHAPES=cfg3,cfg2 EMP_MODES=persist,ablate timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "==|us/iter" | cut -c1-110; } > gpurun_out/r6_em_ab4.log 2>&1
