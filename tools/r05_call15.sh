#!/bin/bash
# round 5, fifteenth GPU session: capsule look-ahead in the mask phase, heading_atan2, LTPL_PIPE1 on the new layer step: parity of the default build,
# same-box A/B with one variant per change
export TMPDIR=/tmp
T=${R05TAG:-r05p}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_edge_mask.py tests/test_fresh_tick_golden.py tests/test_gpu_edge_cases.py tests/test_other_tracks.py tests/test_no_virtual_goal.py tests/test_gpu_wave_ops.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/$T/gputest.txt
LTPL_HIP_LIB=$PWD/$V/pipe1.so timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py -m gpu -x -q > gpurun_out/$T/gputest_pipe1.txt 2>&1; echo "pipe1 tests rc=$?"; tail -1 gpurun_out/$T/gputest_pipe1.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/pipe1.so $V/noahead.so $V/libm.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
