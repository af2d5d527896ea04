#!/bin/bash
# A/B of library variants on ONE box (boxes differ by +-3 %): tools/ab_bench.sh <lib or "base"> ...   -> ticks/s, path kernel ms per variant
# alternating runs, 2 rounds
ARGS="--steps 100 --warmup 10 --no-cpu --latency-ticks 0 --dropin-ticks 0 --no-extra"
for round in 1 2; do
  for V in "$@"; do
    if [ "$V" = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$V; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-32s %.2f M ticks/s  k_paths live %.4f ms  alone %.4f ms  prep %.3f lanes %.3f' % ('$V', d['value'] / 1e6, r['kernel_ms'], r['kernel_ms_not_overlapped'], r['pipeline_ms']['k_follow_prep'], r['pipeline_ms']['k_vel_lanes']))"
  done
done
