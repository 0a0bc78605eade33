#!/bin/bash
# round 6: what a step of the persistent loop is made of now: product, no-wait gate, stream from L1 (L2HIT), VBEM with EM's x (VBCHEAP), phase stamps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "--- product (persist, ablate = no waits)"; EMP_MODES=persist,ablate timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "==|us/iter" | cut -c1-150
for name in "$@"; do echo "--- $name"
  SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_$name.so EMP_MODES=persist timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "us/iter|persist stamps|  tile" | awk '!seen[substr($0,1,60)]++' | cut -c1-400
done
} > gpurun_out/r6_em_ab2.log 2>&1
