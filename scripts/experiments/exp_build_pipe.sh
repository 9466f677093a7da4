#!/bin/bash
# usage: scripts/experiments/exp_build_pipe.sh <name> <file-stem> <flags...> : builds build/exp_<name>/liblmc.so with lmc_<stem>.hip recompiled with the given flags
# (flags replace the default "-fno-slp-vectorize")
name=$1; stem=$2; shift; shift
d=build/exp_$name; mkdir -p $d; rm -f $d/*.o
for f in build/obj/*.o; do [ "$(basename $f)" != $stem.o ] && cp $f $d/; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics "$@" -Iinclude -Ilmc_atomi_amd/csrc -c lmc_atomi_amd/csrc/$stem.hip -o $d/$stem.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "  VGPRs:|ScratchSize" | sed 's/.*remark: *//; s/ \[-R.*//' | sort | uniq -c
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblmc.so $d/*.o
