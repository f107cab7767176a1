#!/bin/bash
# usage: scripts/build_variant.sh NAME "-DFLAG=1 ..."   -> navier_stokes_solver_amd/libnsk_hip_NAME.so
# A study build of the library with other compile-time flags for nsk_kernels.hip (the other objects are reused);
# load it with NSK_HIP_LIBRARY=<path> (scripts/ab_libs.sh, bench.py).
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../navier_stokes_solver_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fopenmp -I../../include $flags -c nsk_kernels.hip -o build/nsk_kernels_$name.o
objs=$(ls build/*.o | grep -v "nsk_kernels" | tr '\n' ' ')
hipcc --offload-arch=gfx950 -fPIC -fopenmp -shared $objs build/nsk_kernels_$name.o -o ../libnsk_hip_$name.so -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib
echo built ../libnsk_hip_$name.so
