#!/bin/bash
# Same-box A/B of ENVIRONMENT settings on the headline (alternating short bench runs, 2 rounds):  tools/env_ab.sh "LTPL_FINAL_Y=8" "LTPL_FINAL_Y=2" ...
ARGS="--steps 100 --warmup 10 --no-cpu --latency-ticks 0 --dropin-ticks 0 --no-extra"
for round in 1 2; do
  for S in "$@"; do
    env $S python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-40s %.2f M ticks/s  k_paths live %.4f ms  alone %.4f ms  prep %.3f lanes %.3f' % ('$S', d['value'] / 1e6, r['kernel_ms'], r['kernel_ms_not_overlapped'], r['pipeline_ms']['k_follow_prep'], r['pipeline_ms']['k_vel_lanes']))"
  done
done
