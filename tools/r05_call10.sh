#!/bin/bash
# round 5, tenth GPU session: priority experiments for the velocity stage next to the path kernel -- s_setprio inside the velocity kernels
# (-DLTPL_VEL_PRIO=1 / 3) and the velocity streams at the highest stream priority (LTPL_VEL_STREAM_PRIO=1); same-box alternating A/B
export TMPDIR=/tmp
mkdir -p gpurun_out/r05j
V=$PWD/graphbasedlocaltrajectoryplanner_amd/csrc/variants
ARGS="--steps 100 --warmup 10 --no-cpu --latency-ticks 0 --dropin-ticks 0 --no-extra"
run() {  # label, lib ('' = default), stream prio
  if [ -n "$2" ]; then export LTPL_HIP_LIB=$2; else unset LTPL_HIP_LIB; fi
  LTPL_VEL_STREAM_PRIO=$3 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-28s %.2f M ticks/s  k_paths live %.4f ms  alone %.4f ms  prep %.3f lanes %.3f' % ('$1', d['value'] / 1e6, r['kernel_ms'], r['kernel_ms_not_overlapped'], r['pipeline_ms']['k_follow_prep'], r['pipeline_ms']['k_vel_lanes']))"
}
for rep in 1 2; do
  run "base" "" 0; run "setprio 1" $V/prio1.so 0; run "setprio 3" $V/prio3.so 0; run "stream prio" "" 1; run "setprio 3 + stream prio" $V/prio3.so 1
done > gpurun_out/r05j/prio_ab.txt 2>&1; cat gpurun_out/r05j/prio_ab.txt
