#!/bin/bash
# round 6: stress runs of the product library against the CPU restatement: EM (random shapes, EM / VBEM, persistent and fallback loops),
# the class builder (random streams, batch cuts, sub-batch sizes), bias-aware effective lengths
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for seed in 201 202 203 204 205 206 207 208; do
  timeout 200 python tools/em_stress.py $seed ${EM_S:-60} 2>&1 | tail -2
done
for seed in 211 212 213 214; do
  timeout 300 python tools/builder_stress.py $seed ${EQ_S:-60} 2>&1 | tail -1 | cut -c1-200
done
timeout 300 python tools/bias_stress.py 221 ${BIAS_S:-45} 2>&1 | tail -1 | cut -c1-200
} > gpurun_out/r6_stress.log 2>&1
