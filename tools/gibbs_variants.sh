#!/bin/bash
# dev tooling: variants of libsfgpu.so that differ in the compile flags of gibbs.hip / sampling.hip (rng.h experiments)
#   tools/gibbs_variants.sh name1:"-DFOO=1" ...   -> sailfish_amd/csrc/variants/libsfgpu_<name>.so
set -e
cd "$(dirname "$0")/../sailfish_amd/csrc"
mkdir -p variants
make -s -j8 all
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-result -DSFGPU_VARIANTS"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  for f in gibbs sampling; do hipcc $FLAGS $defs -c $f.hip -o variants/${f}_$name.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsfgpu_$name.so build/core.o build/eqclass.o build/em.o build/misc.o build/primitives.o \
        variants/sampling_$name.o variants/gibbs_$name.o build/filter.o build/bias.o build/merge.o build/mapper.o build/comm.o -Wl,-rpath,/opt/rocm/lib
  echo built $name "($defs)"
done
