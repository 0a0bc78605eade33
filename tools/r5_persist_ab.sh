#!/bin/bash
# dev (round 5): A/B of build variants of the persistent EM loop on the box:  tools/r5_persist_ab.sh name:"-DFLAG" ...
cd "$(dirname "$0")/.."
tools/em_variants.sh "$@" > /dev/null 2>&1
echo "--- product"; EMP_MODES=${EMP_MODES:-persist} timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "==|us/iter" | cut -c1-120
for spec in "$@"; do name="${spec%%:*}"; echo "--- $name (${spec#*:})"
  SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_$name.so EMP_MODES=${EMP_MODES:-persist} timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "us/iter" | cut -c1-120
done
