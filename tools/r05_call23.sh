#!/bin/bash
# round 5, twenty-third GPU session: position / vehicle data of phases 2 and 3 requested in phase 1 (one-wave form; tools/experiments/r05_early_pos.patch
# was the default build of this session, variant noearly = the loads at their use = the committed source): parity, same-box A/B (C2, C3)
export TMPDIR=/tmp
T=${R05TAG:-r05y}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_edge_mask.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/$T/gputest.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/noearly.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
for L in base $V/noearly.so; do
  if [ "$L" = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$L; fi
  echo "c3 $L: $(timeout 300 python tools/c3_rate.py 32768 2>/dev/null | tail -1 | cut -c1-140)"
done > gpurun_out/$T/c3.txt 2>&1; cat gpurun_out/$T/c3.txt; unset LTPL_HIP_LIB
