#!/bin/bash
# Builds an A/B partner of the product library with extra -D flags on ONE translation unit (the rest is linked from csrc/build/):
#   tools/build_variant.sh <name> <unit.hip> -DFOO=1 ...   ->  gpurun_tmp/libfibers_hip_<name>.so   (load it with FIBERS_HIP_LIB)
set -eu
name=$1; unit=$2; shift 2
cd "$(dirname "$0")/../fibers.jl_amd/csrc"
make -s -j4
mkdir -p ../../gpurun_tmp/var_$name
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result "$@" -c $unit -o ../../gpurun_tmp/var_$name/${unit%.hip}.o
objs=$(ls build/*.o | grep -v "build/${unit%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_tmp/libfibers_hip_$name.so $objs ../../gpurun_tmp/var_$name/${unit%.hip}.o
ls -la ../../gpurun_tmp/libfibers_hip_$name.so
