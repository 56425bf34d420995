#!/bin/bash
# usage: tools/abbuild.sh <name> [-DDEFINE ...]   builds build/ab/<name>.so from the current tree (A/B measurements: GEAR_HIP_LIB=)
name=$1; shift
cd "$(dirname "$0")/.." && mkdir -p build/ab &&
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-slp-vectorize -Wno-unused-function "$@" -o build/ab/$name.so gear_amd/csrc/gear_hip.hip && echo built build/ab/$name.so
