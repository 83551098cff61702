#!/bin/bash
# A copy of the library with other compile-time settings in all three device translation units, for A/B runs:
#   tools/build_variant_all.sh NAME -DPT_REC_STRIDE=4 ...   ->  variants/libmi_pathtrace_NAME.so   (use with MI_PATHTRACE_LIB)
set -e
cd "$(dirname "$0")/../cuda-path-tracer_amd/csrc"
NAME=$1; shift
OUT=../../variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -fno-slp-vectorize"
for f in pt_kernels pt_denoise pt_api; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $OUT/${f}_$NAME.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmi_pathtrace_$NAME.so host/error.o host/scene_host.o host/gltf_load.o host/image_io.o host/bvh_build.o host/pathtrace_shim.o $OUT/pt_kernels_$NAME.o $OUT/pt_denoise_$NAME.o $OUT/pt_api_$NAME.o
rm $OUT/*_$NAME.o
echo built $OUT/libmi_pathtrace_$NAME.so
