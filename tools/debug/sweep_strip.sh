#!/bin/bash
for sz in "40000 10000" "59800 2990" "20000 4000" "5000 5000" "10000 2000" "100000 10000"; do
  set -- $sz
  for m in 0 1; do
    NT=$1 NV=$2 timeout 200 python tools/debug/time_strip.py fp16 $m 2>&1 | grep ms
    NT=$1 NV=$2 timeout 200 python tools/debug/time_strip.py fp16 $m scores 2>&1 | grep ms
  done
done
for m in 0 1; do timeout 200 python tools/debug/time_strip.py bf16 $m 2>&1 | grep ms; timeout 200 python tools/debug/time_strip.py bf16 $m scores 2>&1 | grep ms; done
