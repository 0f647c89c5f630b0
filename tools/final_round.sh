#!/bin/bash
# the round's evidence in one GPU session: profile round (trace + PMC passes + bench line), the other configs' lines, hipBLASLt table
bash tools/profile_round.sh r6 > gpurun_out/prof_r6.log 2>&1
for w in c1_test3k c2_10kx3k c3_framelaff_10kx3k; do
  timeout 600 python bench.py --workload $w 2>/dev/null | tail -1 > gpurun_out/prof_r6/summary/r6_bench_line_$w.json
done
timeout 900 python bench.py --workload c5_ml_100kx30k --precision bf16 --steps 10 2>/dev/null | tail -1 > gpurun_out/prof_r6/summary/r6_bench_line_c5_bf16.json
for sh in text video; do
  timeout 600 python bench.py --emulate-shard 8 --shard $sh 2>/dev/null | tail -1 > gpurun_out/prof_r6/summary/r6_bench_line_c4_shard8_$sh.json
done
timeout 600 python tools/gemm_vs_hipblaslt.py > gpurun_out/prof_r6/summary/r6_gemm_vs_hipblaslt.txt 2>&1
timeout 300 python tools/pcie_inclusive.py > gpurun_out/prof_r6/summary/r6_pcie_inclusive.txt 2>&1
timeout 300 python tools/predict_wall.py > gpurun_out/prof_r6/summary/r6_predict_wall.txt 2>&1
timeout 600 python tools/energy_table.py --out gpurun_out/prof_r6/summary/r6_energy.json > gpurun_out/prof_r6/summary/r6_energy_table.txt 2>/dev/null
timeout 600 python tools/energy_table.py --no-scores --out gpurun_out/prof_r6/summary/r6_energy_noscores.json >> gpurun_out/prof_r6/summary/r6_energy_table.txt 2>/dev/null
ls gpurun_out/prof_r6/summary/
