#!/bin/bash
# round 5, eighth GPU session: LTPL_PIPE1 (one layer of software pipelining in the single-filter sweeps): parity of the variant, same-box A/B,
# SQ counters; mask-phase positions per pass (LTPL_MQ) re-checked with the capsule prefetch
export TMPDIR=/tmp
mkdir -p gpurun_out/r05h
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
LTPL_HIP_LIB=$PWD/$V/pipe1.so timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_edge_mask.py tests/test_fresh_tick_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/r05h/gputest_pipe1.txt 2>&1; echo "pipe1 tests rc=$?"; tail -3 gpurun_out/r05h/gputest_pipe1.txt
timeout 900 tools/ab_bench.sh base $V/pipe1.so $V/mq3.so $V/mq1.so > gpurun_out/r05h/ab_bench.txt 2>&1; cat gpurun_out/r05h/ab_bench.txt
PMC_TRAFFIC=0 PMC_TAG=r05h timeout 600 tools/pmc_ab.sh $V/pipe1.so > gpurun_out/r05h/pmc_pipe1.txt 2>&1; cat gpurun_out/r05h/pmc_pipe1.txt
