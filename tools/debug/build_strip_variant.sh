#!/bin/bash
# variant builds of the strip GEMM: tools/debug/build_strip_variant.sh NAME -DFLAG...  -> scratch/strip_NAME/liblaff_hip.so
set -e
cd "$(dirname "$0")/../.."
n=$1; shift
mkdir -p scratch/strip_$n
for o in laff_amd/lib/*.o; do [ "$(basename $o)" = sim_strip.o ] || cp $o scratch/strip_$n/; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc "$@" -c laff_amd/csrc/sim_strip.hip -o scratch/strip_$n/sim_strip.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/strip_$n/liblaff_hip.so scratch/strip_$n/*.o
