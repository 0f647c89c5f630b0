#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root, e.g. through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> profiles/<tag>_bench_kernel_trace.txt
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)             -> profiles/<tag>_bench_pmc_{fetch,write}.txt, <tag>_traffic.json
#   3. the un-profiled bench line                                     -> profiles/<tag>_bench_line.json
# Counter passes never combine --pmc with a trace domain (MI355X_MICROARCH.md; the pool refuses that combination).
set -e
TAG=${1:-r6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG        # summaries + logs (copied back by gpurun: <= 64 MiB)
RAW=/tmp/prof_raw_$TAG                # rocprofv3's own output trees stay on the box
mkdir -p "$OUT" "$RAW" "$ROOT/profiles"
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --profile-steps 5 --no-cpu-baseline --no-extra-modes"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$RAW/trace" -o t --output-format csv -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$RAW/fetch" -o f --output-format csv -- $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$RAW/write" -o w --output-format csv -- $BENCH > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d "$RAW/mfma" -o m --output-format csv -- $BENCH > "$OUT/mfma.log" 2>&1
cd "$ROOT"
# summaries go to gpurun_out/ (the only directory gpurun copies back); `cp gpurun_out/prof_<tag>/summary/* profiles/` commits them
S=$OUT/summary
mkdir -p "$S"
python3 tools/rocprof_summary.py "$RAW/trace" --out $S/${TAG}_bench_kernel_trace.txt
python3 tools/rocprof_summary.py "$RAW/fetch" --out $S/${TAG}_bench_pmc_fetch.txt
python3 tools/rocprof_summary.py "$RAW/write" --out $S/${TAG}_bench_pmc_write.txt
python3 tools/mfma_util.py "$RAW/mfma" > $S/${TAG}_bench_pmc_mfma.txt
python3 tools/make_traffic.py "$RAW/fetch" "$RAW/write" $S/${TAG}_traffic.json > /dev/null
cp $S/${TAG}_traffic.json profiles/${TAG}_traffic.json          # bench.py reads roofline.traffic from here
python3 tools/timeline.py "$RAW/trace" > $S/${TAG}_bench_timeline.txt || true
cp "$RAW/trace/t_kernel_stats.csv" $S/${TAG}_bench_kernel_stats.csv
grep '"metric"' "$OUT/trace.log" | tail -1 > $S/${TAG}_bench_line_under_rocprof.json || true   # the bench's own HIP-event durations in the traced run
python3 bench.py --steps 20 --warmup 5 > $S/${TAG}_bench_line.json 2> "$OUT/bench.err"
tail -c 700 $S/${TAG}_bench_line.json
