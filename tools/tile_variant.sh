#!/bin/bash
# tools/tile_variant.sh <name> <-Dflags...>: rebuild only seg_tile.hip with the flags and link against the main build's other objects
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/pygho_amd/_lib/variants/$name; mkdir -p $V
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -fno-gpu-rdc -DNDEBUG "$@" -I$R/include -c $R/pygho_amd/csrc/seg_tile.hip -o $V/seg_tile.hip.o || exit 1
objs=$(ls $R/pygho_amd/_lib/obj/*.hip.o | grep -v seg_tile)
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs $V/seg_tile.hip.o -o $V/libpygho_hip.so && rm $V/seg_tile.hip.o && echo built $V
