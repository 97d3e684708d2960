#!/bin/bash
# usage: tools/build_variant.sh <name> [-DNB_...=..]...   -> build/lib_<name>.so (experimental builds for tools/variants.sh)
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" fluidsimulation_amd/csrc/sphfluid.hip -o build/lib_$name.so
