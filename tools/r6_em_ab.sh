#!/bin/bash
# round 6: the persistent loop's new class records against the round-5 kernel (variants/libsfgpu_base.so), tests first
#   tools/r6_em_ab.sh [variant names built by tools/em_variants.sh ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "=== tests skipped"
echo "--- product"; EMP_MODES=persist timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "==|us/iter" | cut -c1-150
for name in "$@"; do echo "--- $name"
  SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_$name.so EMP_MODES=persist timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "us/iter" | cut -c1-150
done
} > gpurun_out/r6_em_ab.log 2>&1
