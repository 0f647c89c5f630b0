#!/bin/bash
# debug build of the library with extra -D flags: build_variant.sh <name> <flags...>  -> scratch/var_<name>/liblaff_hip.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p scratch/var_$name
for f in api fuse gemm_nt rank loss; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc "$@" -Iinclude -c laff_amd/csrc/$f.hip -o scratch/var_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/var_$name/liblaff_hip.so scratch/var_$name/*.o
