#!/bin/bash
# debug build: gemm_nt.hip + api.hip with -DLAFF_GEMM_TRACE (s_memtime probes per tile, wait breakdown of the K loop, epilogue stamps),
# every other object as last built by `python -m laff_amd.build`  ->  scratch/gtrace/liblaff_hip.so  (load with LAFF_HIP_LIB)
set -e
cd "$(dirname "$0")/../.."
mkdir -p scratch/gtrace
for o in laff_amd/lib/*.o; do b=$(basename $o); [ $b = gemm_nt.o ] || [ $b = api.o ] || cp $o scratch/gtrace/; done
for f in gemm_nt api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -DLAFF_GEMM_TRACE "$@" -Iinclude -c laff_amd/csrc/$f.hip -o scratch/gtrace/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/gtrace/liblaff_hip.so scratch/gtrace/*.o
