#!/bin/bash
# round 5, third GPU session: closest-layer grid + pipelined capsule loads: parity, same-box A/B (C2 and C3), fleet size sweep, tick phases
export TMPDIR=/tmp
mkdir -p gpurun_out/r05c
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05c/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r05c/gputest.txt
timeout 600 tools/ab_bench.sh $V/dpp.so base > gpurun_out/r05c/ab_bench.txt 2>&1; cat gpurun_out/r05c/ab_bench.txt
for rep in 1 2; do
  LTPL_HIP_LIB=$PWD/$V/dpp.so timeout 200 python tools/c3_rate.py 8192 32768 2>/dev/null | grep "^c3" >> gpurun_out/r05c/c3_ab.txt
  timeout 200 python tools/c3_rate.py 8192 32768 2>/dev/null | grep "^c3" >> gpurun_out/r05c/c3_ab.txt
done; cat gpurun_out/r05c/c3_ab.txt
for N in 8192 16384 32768; do timeout 300 python tools/fleet_rate.py --planners $N --ticks 200 --mix 2>&1 | tail -3 >> gpurun_out/r05c/fleet_sizes.txt; done; cat gpurun_out/r05c/fleet_sizes.txt
timeout 200 python tools/dbg_tick_timing.py > gpurun_out/r05c/tick_phases.txt 2>&1; grep "ltpl dbg" gpurun_out/r05c/tick_phases.txt | tail -12
