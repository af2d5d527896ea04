#!/bin/bash
# One GPU-box session of a build round (run through gpurun): parity tests, bench line, kernel stats, HBM traffic and lane
# utilisation counters. Everything lands under gpurun_out/<tag>_*; tools/profile_summarise.py copies the summaries to profiles/.
#   tools/gpu_round.sh <tag> [skip-tests]
set -u
TAG=${1:-r02a}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "${2:-}" != "skip-tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1
  echo "gpu tests rc=$?"; tail -5 gpurun_out/${TAG}_gputest.log
fi
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; cat gpurun_out/${TAG}_bench.json | head -c 6000; tail -3 gpurun_out/${TAG}_bench.err
tools/profile_round.sh ${TAG}
tools/pmc_pass.sh ${TAG}_lanes "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" > gpurun_out/${TAG}_lanes.txt 2>&1
tools/pmc_pass.sh ${TAG}_occ "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" > gpurun_out/${TAG}_occ.txt 2>&1
cat gpurun_out/${TAG}_lanes.txt gpurun_out/${TAG}_occ.txt | head -80
