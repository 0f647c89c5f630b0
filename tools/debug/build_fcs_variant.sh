#!/bin/bash
# variant builds of the strip FC: tools/debug/build_fcs_variant.sh NAME -DFLAG...  -> scratch/fcs_NAME/liblaff_hip.so  (load with LAFF_HIP_LIB)
set -e
cd "$(dirname "$0")/../.."
n=$1; shift
mkdir -p scratch/fcs_$n
for o in laff_amd/lib/*.o; do [ "$(basename $o)" = fc_strip.o ] || cp $o scratch/fcs_$n/; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc "$@" -c laff_amd/csrc/fc_strip.hip -o scratch/fcs_$n/fc_strip.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/fcs_$n/liblaff_hip.so scratch/fcs_$n/*.o
