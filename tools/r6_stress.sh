#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for seed in 101 102 103 104 105 106; do
  timeout 200 python tools/em_stress.py $seed 45 2>&1 | tail -2
done
} > gpurun_out/r6_stress.log 2>&1
