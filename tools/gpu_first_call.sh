#!/bin/bash
# First gpurun call of a round (one box acquisition for the checks that need hardware but no profiler):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_first_call.sh'
# 1. the regular GPU suite + smoke (must stay green)
#    (since round 3 the suite includes the GPU legs of the additional tracks: zalazone, millbrook, lvms, berlin, modena)
# 2. the planner / edge-case GPU tests through a build of the library with libstdc++ assertions in its host code
#    (-D_GLIBCXX_ASSERTIONS: vector bounds on REAL kernel results; the sanitizer runs of tools/fakehip only see empty results)
# 3. closed-loop rate of 256 planners with 1 / 4 / 8 host threads
# Logs under gpurun_out/first_call/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/first_call; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "gpu suite: exit $?" | tee $OUT/summary.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke: exit $?" | tee -a $OUT/summary.txt
mkdir -p /tmp/ltpl_assert
if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -D_GLIBCXX_ASSERTIONS -Wno-unused-function \
     -o /tmp/ltpl_assert/libltpl_hip.so graphbasedlocaltrajectoryplanner_amd/csrc/ltpl_hip.hip > $OUT/assert_build.log 2>&1; then
  LTPL_HIP_LIB=/tmp/ltpl_assert/libltpl_hip.so timeout 600 python -m pytest tests/test_gpu_planner.py tests/test_gpu_edge_cases.py tests/test_gpu_vel.py \
     -m gpu -q > $OUT/assert_tests.log 2>&1
  echo "host assertions build: tests exit $?" | tee -a $OUT/summary.txt
  tail -2 $OUT/assert_tests.log | tee -a $OUT/summary.txt
else
  echo "host assertions build: compile failed (see assert_build.log)" | tee -a $OUT/summary.txt
fi
# 3. a batch of host planners (ltpl_planner_*)
timeout 300 python tools/planner_batch_rate.py --planners 256 --ticks 200 2>&1 | tail -1 | tee -a $OUT/summary.txt
