#!/bin/bash
# fp32 vs fp64 operand records of the lane kernels (|kappa|, element length): ticks/s and worst relative error per quantity of both builds on
# ONE box.   tools/vel_precision_ab.sh <fp32 variant library>     (round 6: fp64 records are the default; the variant is built with -DLTPL_VEL_F32_OPERANDS)
ARGS="--steps 100 --warmup 10 --latency-ticks 0 --dropin-ticks 0 --no-extra --cpu-sample 1024"
for round in 1 2; do
  for V in base "$1"; do
    if [ "$V" = base ]; then unset LTPL_HIP_LIB; T="fp64 operands (default)"; else export LTPL_HIP_LIB=$PWD/$V; T="fp32 operands ($V)"; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; p = d['parity_detail']
print('%-40s %.2f M ticks/s  lanes %.3f ms prep %.3f ms  parity %s  max_rel_err %.2e  vx %.2e ax %.2e kappa %.2e' % ('$T', d['value'] / 1e6,
      r['pipeline_ms']['k_vel_lanes'], r['pipeline_ms']['k_follow_prep'], d['parity_checked'], p['max_rel_err'], p['max_rel_err_by_quantity']['vx'],
      p['max_rel_err_by_quantity']['ax'], p['max_rel_err_by_quantity']['kappa']))"
  done
done
