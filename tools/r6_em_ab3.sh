#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for name in "$@"; do echo "--- $name"
  SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_$name.so EMP_SHAPES=${EMP_SHAPES:-cfg3} EMP_MODES=persist timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "us/iter|persist stamps|  tile" | awk '!seen[substr($0,1,60)]++' | cut -c1-330
done
} > gpurun_out/r6_em_ab3.log 2>&1
