#!/bin/bash
# ablation builds of the GEMM (wrong results, timing only): scratch/abl_<name>/liblaff_hip.so
set -e
cd "$(dirname "$0")/../.."
for abl in NODMA NOREAD; do
  d=scratch/abl_$abl; mkdir -p $d
  for f in api fuse rank loss; do cp laff_amd/lib/$f.o $d/$f.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -DLAFF_ABL_$abl -Iinclude -c laff_amd/csrc/gemm_nt.hip -o $d/gemm_nt.o &
done
wait
for abl in NODMA NOREAD; do d=scratch/abl_$abl; /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblaff_hip.so $d/*.o; done
