#!/bin/bash
# variant build of ONE source of the library: tools/debug/build_one_variant.sh NAME FILE -DFLAG...  -> scratch/v_NAME/liblaff_hip.so
# (the other objects are the ones of the last `python -m laff_amd.build`; load with LAFF_HIP_LIB)
set -e
cd "$(dirname "$0")/../.."
n=$1; f=$2; shift; shift
mkdir -p scratch/v_$n
for o in laff_amd/lib/*.o; do [ "$(basename $o)" = $f.o ] || cp $o scratch/v_$n/; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -Iinclude "$@" -c laff_amd/csrc/$f.hip -o scratch/v_$n/$f.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/v_$n/liblaff_hip.so scratch/v_$n/*.o
