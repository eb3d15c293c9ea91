#!/bin/bash
# Experiment builds of the device library: tools/build_variant.sh <name> [-DFLAG ...] -> mygram-db_amd/libmygram_gpu_<name>.so
# (git-ignored; select with MGX_LIBRARY=<path> — engine.py / the replay leg of bench.py only, the shim links the product library)
set -e
NAME=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude $*"
C=mygram-db_amd/csrc
/opt/rocm/bin/hipcc $F -c $C/mgx_kernels.hip -o /tmp/mgx_kernels_$NAME.o
/opt/rocm/bin/hipcc $F -x hip -c $C/mgx_api.cpp -o /tmp/mgx_api_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mygram-db_amd/libmygram_gpu_$NAME.so /tmp/mgx_kernels_$NAME.o /tmp/mgx_api_$NAME.o $C/mgx_columns.o $C/mgx_tools.o -pthread
echo built mygram-db_amd/libmygram_gpu_$NAME.so
