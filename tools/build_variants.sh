#!/bin/bash
# dev tooling: build tuning variants of libsfgpu.so with different sweep geometry into gpurun_out-free paths
set -e
cd "$(dirname "$0")/../sailfish_amd/csrc"
mkdir -p variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -ffp-contract=off"
for cfg in "$@"; do
  IFS=_ read -r T B U <<< "$cfg"
  hipcc $FLAGS -DSFGPU_TILE_NNZ=$T -DSFGPU_SWEEP_BLOCK=$B -DSFGPU_SWEEP_UNROLL=$U -c em.hip -o variants/em_$cfg.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsfgpu_$cfg.so build/core.o build/eqclass.o build/misc.o build/primitives.o variants/em_$cfg.o -Wl,-rpath,/opt/rocm/lib
  echo built $cfg
done
