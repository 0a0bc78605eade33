#!/bin/bash
# dev tooling: build variants of libsfgpu.so that differ in eqclass.hip compile flags.
#   tools/eq_variants.sh name1:"-DFOO -DBAR=2" name2:"-DBAZ" ...   -> sailfish_amd/csrc/variants/libsfgpu_<name>.so
set -e
cd "$(dirname "$0")/../sailfish_amd/csrc"
mkdir -p variants
make -s -j8 all
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-result -DSFGPU_VARIANTS"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  hipcc $FLAGS $defs -c eqclass.hip -o variants/eqclass_$name.o
  hipcc $FLAGS -c core.hip -o variants/core_v.o           # (sfgpu_has_variants() answers 1)
  hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsfgpu_$name.so variants/core_v.o variants/eqclass_$name.o build/em.o build/misc.o build/primitives.o \
        build/sampling.o build/gibbs.o build/filter.o build/bias.o build/merge.o build/mapper.o build/comm.o -Wl,-rpath,/opt/rocm/lib
  echo built $name "($defs)"
done
