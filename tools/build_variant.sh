#!/bin/bash
# Builds one library variant for tools/sweep.sh: gpurun_variants/<name>.so from the current sources with extra
# compiler flags (objects go to a scratch directory, the in-tree build is left alone).
# usage: tools/build_variant.sh <name> [-DKNOB=value ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/path_tracer_amd/csrc
tmp=$(mktemp -d)
mkdir -p "$root/gpurun_variants"
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DMIPT_TRAV_WAVES=5 -DMIPT_LEAN_WAVES=6 $*"
/opt/rocm/bin/hipcc $flags -c "$src/kernels.hip" -o "$tmp/kernels.o" &
/opt/rocm/bin/hipcc $flags -c "$src/gpu_build.hip" -o "$tmp/gpu_build.o" &
/opt/rocm/bin/hipcc $flags -x hip -c "$src/api.cpp" -o "$tmp/api.o" &
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -Wall -c "$src/scene_build.cpp" -o "$tmp/scene_build.o" &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$root/gpurun_variants/$name.so" "$tmp"/kernels.o "$tmp"/api.o "$tmp"/scene_build.o "$tmp"/gpu_build.o -lpthread
rm -rf "$tmp"
echo "built gpurun_variants/$name.so"
