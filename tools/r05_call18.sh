#!/bin/bash
# round 5, eighteenth GPU session: tail edges without control flow in rounds 0 / 1 (cost and edge word requested together), unconditional election
# add; variant: the first 64 tail edges prefetched at the top of the layer step in every plan class. Parity, same-box A/B (C2, C3)
export TMPDIR=/tmp
T=${R05TAG:-r05s}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/$T/gputest.txt
LTPL_HIP_LIB=$PWD/$V/tpfall.so timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_other_tracks.py -m gpu -x -q > gpurun_out/$T/gputest_tpfall.txt 2>&1; echo "tpfall tests rc=$?"; tail -1 gpurun_out/$T/gputest_tpfall.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/tpfall.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
for L in base $V/r05i.so $V/tpfall.so; do
  if [ "$L" = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$L; fi
  echo "c3 $L: $(timeout 300 python tools/c3_rate.py 32768 2>/dev/null | tail -1 | cut -c1-140)"
done > gpurun_out/$T/c3.txt 2>&1; cat gpurun_out/$T/c3.txt; unset LTPL_HIP_LIB
