#!/bin/bash
# tools/variant.sh <source stem> <name> <-Dflags...>: rebuild only csrc/<stem>.hip with the flags and link it against the main
# build's other objects into pygho_amd/_lib/variants/<name>/libpygho_hip.so (run with PYGHO_AMD_LIB=<that file>)
stem=$1; name=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/pygho_amd/_lib/variants/$name; mkdir -p $V
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -fno-gpu-rdc -DNDEBUG "$@" -I$R/include -c $R/pygho_amd/csrc/$stem.hip -o $V/$stem.hip.o || exit 1
objs=$(ls $R/pygho_amd/_lib/obj/*.hip.o | grep -v "/$stem.hip.o")
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs $V/$stem.hip.o -o $V/libpygho_hip.so && rm $V/$stem.hip.o && echo built $V
