#!/bin/bash
# dev (round 5): cache policy of the route pass's granule stores, per-kernel times under rocprofv3 (cfg3)
cd "$(dirname "$0")/.."
tools/eq_variants.sh nt:"-DSFGPU_X_STORE_POLICY=1" sc1:"-DSFGPU_X_STORE_POLICY=2" sc01:"-DSFGPU_X_STORE_POLICY=3" sc1nt:"-DSFGPU_X_STORE_POLICY=4" > /dev/null 2>&1
bash tools/eq_kernels.sh main nt sc1 sc01 sc1nt 2>&1 | grep -E "==|k_part_route|k_part_insert" | cut -c1-110
