#!/bin/bash
# round 5, twentieth GPU session: heading_sincos for the start heading (variant libmsc = the library routine), node step read by every lane as a
# variant: parity, same-box A/B, phase stamps
export TMPDIR=/tmp
T=${R05TAG:-r05u}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py tests/test_gpu_wave_ops.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/$T/gputest.txt
LTPL_HIP_LIB=$PWD/$V/nvsel.so timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py -m gpu -x -q > gpurun_out/$T/gputest_nvsel.txt 2>&1; echo "nvsel tests rc=$?"; tail -1 gpurun_out/$T/gputest_nvsel.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/libmsc.so $V/nvsel.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
timeout 300 python tools/dbg_paths_timing.py 32768 c2 > gpurun_out/$T/phases_c2.txt 2>&1; grep "ltpl dbg" gpurun_out/$T/phases_c2.txt | tail -5 | head -2
