#!/bin/bash
# ablation builds of the strip GEMM (timing only, wrong results): scratch/strip_abl_<n>/liblaff_hip.so
set -e
cd "$(dirname "$0")/../.."
for n in "$@"; do
  mkdir -p scratch/strip_abl_$n
  cp laff_amd/lib/api.o laff_amd/lib/fuse.o laff_amd/lib/gemm_nt.o laff_amd/lib/rank.o laff_amd/lib/loss.o scratch/strip_abl_$n/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -DLAFF_STRIP_ABL=$n -c laff_amd/csrc/sim_strip.hip -o scratch/strip_abl_$n/sim_strip.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/strip_abl_$n/liblaff_hip.so scratch/strip_abl_$n/*.o
done
