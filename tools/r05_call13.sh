#!/bin/bash
# round 5, thirteenth GPU session: layer step of the sweeps -- chunk loads addressed as scalar base + lane offset (sentinel select at consumption
# through a scalar lane mask), the layer table row carried from the prefetch, scalar discount / chunk / tail tests: parity, same-box A/B against the
# r05i build, SQ counters
export TMPDIR=/tmp
mkdir -p gpurun_out/${R05TAG:-r05m}
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_edge_mask.py tests/test_fresh_tick_golden.py tests/test_gpu_edge_cases.py tests/test_other_tracks.py tests/test_no_virtual_goal.py -m gpu -x -q > gpurun_out/${R05TAG:-r05m}/gputest.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${R05TAG:-r05m}/gputest.txt
timeout 600 tools/ab_bench.sh base $V/r05i.so > gpurun_out/${R05TAG:-r05m}/ab_bench.txt 2>&1; cat gpurun_out/${R05TAG:-r05m}/ab_bench.txt
PMC_TRAFFIC=0 PMC_TAG=r05m timeout 600 tools/pmc_ab.sh base $V/r05i.so > gpurun_out/${R05TAG:-r05m}/pmc.txt 2>&1; cat gpurun_out/${R05TAG:-r05m}/pmc.txt
