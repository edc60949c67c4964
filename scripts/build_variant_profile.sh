#!/bin/bash
# As build_variant.sh, for variants of the PROFILE library (timing ablations / MST_INJECT pricing experiments):
#   scripts/build_variant_profile.sh <name> "<extra -D flags for mst_scale_space.hip>"   -> scratch_libs/<name>.so
set -e
cd "$(dirname "$0")/../mustache_amd/csrc"
name=$1; shift
flags="$*"
make -j8 PROFILE=1 >/dev/null
mkdir -p ../../scratch_libs build_var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    -fno-honor-nans -mno-amdgpu-ieee -DMST_PROFILE $flags -c mst_scale_space.hip -o build_var_$name/ss.o
objs=$(ls build_profile/*.o | grep -v mst_scale_space)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch_libs/$name.so $objs build_var_$name/ss.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib
python ../../scripts/kernel_resources.py ../../scratch_libs/$name.so "scale_space_kernel" | grep "Tile<" | grep -v "28, 4" | head -4
