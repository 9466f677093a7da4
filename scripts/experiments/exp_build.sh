#!/bin/bash
# usage: scripts/experiments/exp_build.sh <name> <flags...> : builds build/exp_<name>/liblmc.so with extra flags for lmc_step_split.hip (K=10 only)
name=$1; shift
d=build/exp_$name; mkdir -p $d
for f in lmc_capi lmc_ops lmc_step_tile lmc_step_stream; do cp build/obj/$f.o $d/; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -DLMC_NO_STEADY -DLMC_ONLY_K10 "$@" -Iinclude -Ilmc_atomi_amd/csrc -c lmc_atomi_amd/csrc/lmc_step_split.hip -o $d/lmc_step_split.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblmc.so $d/*.o
