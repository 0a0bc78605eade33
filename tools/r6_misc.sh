#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "=== residency probe skipped"
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -6
} > gpurun_out/r6_misc.log 2>&1
