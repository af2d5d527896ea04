#!/bin/bash
# round 5, nineteenth GPU session: the complete GPU suite on the build with initialised edge registers (the experiment build of the previous
# step faulted on the C3 lattice: an indeterminate register chunk was copied / looked at), same-box A/B, C3
export TMPDIR=/tmp
T=${R05TAG:-r05t}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/$T/gputest.txt
timeout 600 tools/ab_bench.sh base $V/r05i.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
for L in base $V/r05i.so; do
  if [ "$L" = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$L; fi
  echo "c3 $L: $(timeout 300 python tools/c3_rate.py 32768 2>/dev/null | tail -1 | cut -c1-140)"
done > gpurun_out/$T/c3.txt 2>&1; cat gpurun_out/$T/c3.txt; unset LTPL_HIP_LIB
