#!/bin/bash
# variant build of SEVERAL sources: tools/debug/build_variant2.sh NAME "file1 file2" -DFLAG...  -> scratch/v_NAME/liblaff_hip.so
set -e
cd "$(dirname "$0")/../.."
n=$1; files=$2; shift; shift
mkdir -p scratch/v_$n
for o in laff_amd/lib/*.o; do cp $o scratch/v_$n/; done
for f in $files; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -Iinclude "$@" -c laff_amd/csrc/$f.hip -o scratch/v_$n/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/v_$n/liblaff_hip.so scratch/v_$n/*.o
