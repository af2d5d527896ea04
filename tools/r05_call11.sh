#!/bin/bash
# round 5, eleventh GPU session: occupancy of the fleet's stage-A kernels (paths_post + vel_a: 126 VGPRs = 4 waves per SIMD, the longest kernel of
# a fleet tick) -- builds for 5 / 6 / 8 waves per SIMD spill to scratch; same-box A/B on the mixed tape, 32 768 planners
export TMPDIR=/tmp
mkdir -p gpurun_out/r05k
V=$PWD/graphbasedlocaltrajectoryplanner_amd/csrc/variants
for rep in 1 2; do
  for L in base fw5 fw6 fw8; do
    if [ $L = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$V/$L.so; fi
    echo "[$L]" >> gpurun_out/r05k/fleet_occ_ab.txt
    timeout 300 python tools/fleet_rate.py --planners 32768 --ticks 100 --mix --reps 1 2>&1 | tail -3 >> gpurun_out/r05k/fleet_occ_ab.txt
  done
done; unset LTPL_HIP_LIB; cat gpurun_out/r05k/fleet_occ_ab.txt
