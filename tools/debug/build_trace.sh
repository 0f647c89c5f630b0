#!/bin/bash
# debug build of the library with s_memtime probes in the GEMM (scratch/tracelib/liblaff_hip.so)
set -e
cd "$(dirname "$0")/../.."
mkdir -p scratch/tracelib
for f in api fuse gemm_nt sim_strip rank loss; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -DLAFF_GEMM_TRACE -DLAFF_STRIP_TRACE -Iinclude -c laff_amd/csrc/$f.hip -o scratch/tracelib/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/tracelib/liblaff_hip.so scratch/tracelib/*.o
