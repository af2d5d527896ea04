#!/bin/bash
# round 5, twenty-first GPU session: shell flush of the obstacle mask with eight entries per pass (variant shell4 = four): parity (mask tests
# included), same-box A/B (C2, C3), phase stamps
export TMPDIR=/tmp
T=${R05TAG:-r05v}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_edge_mask.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/$T/gputest.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/shell4.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
for L in base $V/shell4.so; do
  if [ "$L" = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$L; fi
  echo "c3 $L: $(timeout 300 python tools/c3_rate.py 32768 2>/dev/null | tail -1 | cut -c1-140)"
done > gpurun_out/$T/c3.txt 2>&1; cat gpurun_out/$T/c3.txt; unset LTPL_HIP_LIB
timeout 300 python tools/dbg_paths_timing.py 32768 c2 > gpurun_out/$T/phases_c2.txt 2>&1; grep "ltpl dbg" gpurun_out/$T/phases_c2.txt | tail -5 | head -2
