#!/bin/bash
# round 5, seventeenth GPU session: unconditional election add as a variant: parity, same-box A/B
export TMPDIR=/tmp
T=${R05TAG:-r05r}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
LTPL_HIP_LIB=$PWD/$V/uadd.so timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py -m gpu -x -q > gpurun_out/$T/gputest_uadd.txt 2>&1; echo "uadd tests rc=$?"; tail -1 gpurun_out/$T/gputest_uadd.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/uadd.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
