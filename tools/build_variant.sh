#!/bin/bash
# Experiment helper (CPU): tools/build_variant.sh NAME -DFOO=1 ...  ->  ray-tracing_amd/build/ab/librt_NAME.so, the library with
# rt_kernels.hip compiled under the given defines (loaded through RT_AMD_LIB by tools/ab_*.py).  Not part of the product build.
set -e
cd "$(dirname "$0")/../ray-tracing_amd"
name=$1; shift
mkdir -p build/ab
make -s csrc/rt_api.o csrc/bvh_build.o csrc/rt_tools.o host/host_api.o host/image_io.o host/ray_tracing.o build/linkname/libamdhip64.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -I../include --offload-arch=gfx950 -fno-gpu-rdc "$@" -c ${SRC:-csrc/rt_kernels.hip} -o build/ab/rt_kernels_$name.o 2>/dev/null
g++ -shared -o build/ab/librt_$name.so csrc/rt_api.o csrc/bvh_build.o csrc/rt_tools.o host/host_api.o host/image_io.o host/ray_tracing.o build/ab/rt_kernels_$name.o \
    -Wl,--no-as-needed -Lbuild/linkname -lamdhip64 -Wl,--as-needed -Wl,-rpath,/opt/rocm/lib -lpthread
echo build/ab/librt_$name.so
