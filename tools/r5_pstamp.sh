#!/bin/bash
# dev (round 5): phase stamps of the persistent EM loop -- builds the -DSFGPU_P_STAMP variant on the box and runs the probe with it and with the product library
#   tools/r5_pstamp.sh [shapes] [modes]
cd "$(dirname "$0")/.."
tools/em_variants.sh pstamp:"-DSFGPU_P_STAMP=${PSTAMP:-1}" > /dev/null 2>&1
echo "--- product library"
EMP_SHAPES=${1:-cfg3,cfg2} EMP_MODES=${2:-fused,persist,ablate} timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "==|us/iter"
echo "--- stamp build"
SFGPU_LIB_PATH=$PWD/sailfish_amd/csrc/variants/libsfgpu_pstamp.so EMP_SHAPES=${1:-cfg3,cfg2} EMP_MODES=persist timeout 300 python tools/r5_persist_probe.py 2>&1 | grep -E "==|persist stamps|  tile" | awk '!seen[substr($0,1,40)]++ || /==/'
