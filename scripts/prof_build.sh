#!/bin/bash
# Builds build/librt355_<tag>.so with extra -D flags:  scripts/prof_build.sh <tag> [-DRT_PROFILE ...]
tag=$1; shift
cd "$(dirname "$0")/../raytracing_amd/csrc" && mkdir -p ../../build /tmp/rtb_$tag
for f in rt_kernels.hip rt_scene.cpp rt_api.cpp rt_lbvh.hip rt_lgrid.hip rt_shards.hip rt_sah.cpp; do
  x=""; case $f in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" $x -c $f -o /tmp/rtb_$tag/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/librt355_$tag.so /tmp/rtb_$tag/*.o && echo built build/librt355_$tag.so
