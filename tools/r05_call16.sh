#!/bin/bash
# round 5, sixteenth GPU session: q^(-3/2) for the curvature of the path samples (hardware rsq + Newton), unconditional frontier minimum as a variant:
# parity of the default build and of the variant, same-box A/B
export TMPDIR=/tmp
T=${R05TAG:-r05q}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/$T/gputest.txt
LTPL_HIP_LIB=$PWD/$V/uncond.so timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py -m gpu -x -q > gpurun_out/$T/gputest_uncond.txt 2>&1; echo "uncond tests rc=$?"; tail -1 gpurun_out/$T/gputest_uncond.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/uncond.so $V/ksqrt.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
