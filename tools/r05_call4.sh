#!/bin/bash
# round 5, fourth GPU session: lane-wise follow jobs of the fleet (A/B against the wave-per-job form), prefetched tail edges of plan class B (C3 A/B), parity
export TMPDIR=/tmp
mkdir -p gpurun_out/r05d
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05d/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r05d/gputest.txt
for N in 8192 32768; do
  for S in "LTPL_FLEET_FOLLOW_WAVES=1" "LTPL_FLEET_FOLLOW_WAVES=0"; do
    echo "[$S] planners $N" >> gpurun_out/r05d/fleet_follow_ab.txt
    env $S timeout 300 python tools/fleet_rate.py --planners $N --ticks 200 --mix 2>&1 | tail -3 >> gpurun_out/r05d/fleet_follow_ab.txt
  done
done; cat gpurun_out/r05d/fleet_follow_ab.txt
for rep in 1 2; do
  LTPL_HIP_LIB=$PWD/$V/notpf.so timeout 200 python tools/c3_rate.py 8192 32768 2>/dev/null | grep "^c3" >> gpurun_out/r05d/c3_tail_ab.txt
  timeout 200 python tools/c3_rate.py 8192 32768 2>/dev/null | grep "^c3" >> gpurun_out/r05d/c3_tail_ab.txt
done; cat gpurun_out/r05d/c3_tail_ab.txt
