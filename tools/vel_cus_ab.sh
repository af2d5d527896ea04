#!/bin/bash
# A/B of LTPL_VEL_CUS (velocity streams confined to n compute units by a CU mask) on ONE box: tools/vel_cus_ab.sh <n> ...   (0 = no mask)
ARGS="--steps 100 --warmup 10 --no-cpu --latency-ticks 0 --dropin-ticks 0 --no-extra"
for round in 1 2; do
  for N in "$@"; do
    LTPL_VEL_CUS=$N python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('LTPL_VEL_CUS=%-4s %.2f M ticks/s  ms/step %.4f  k_paths live %.4f ms  alone %.4f ms  parity %s' % ('$N', d['value'] / 1e6, d['ms_per_step'], r['kernel_ms'], r['kernel_ms_not_overlapped'], d.get('parity_checked')))"
  done
done
