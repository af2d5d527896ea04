#!/bin/bash
# round 5, first GPU session: parity of the DPP build, same-box A/B of the variants, hipGraph tick A/B, C3 issue counters
export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r05a/gputest.txt
timeout 900 tools/ab_bench.sh $V/r04.so base $V/vel_8_6.so $V/vel_12_8.so > gpurun_out/r05a/ab_bench.txt 2>&1; cat gpurun_out/r05a/ab_bench.txt
timeout 600 tools/tick_ab.sh "" "LTPL_TICK_GRAPH=1" "LTPL_POLL=1" "LTPL_POLL=1 LTPL_TICK_GRAPH=1" > gpurun_out/r05a/tick_ab.txt 2>&1; cat gpurun_out/r05a/tick_ab.txt
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE|SQC_" | head -60 > gpurun_out/r05a/counters_avail.txt; wc -l gpurun_out/r05a/counters_avail.txt
PMC_TRAFFIC=0 PMC_WORKLOAD=c3 PMC_TAG=r05a timeout 600 tools/pmc_ab.sh base > gpurun_out/r05a/pmc_c3.txt 2>&1; cat gpurun_out/r05a/pmc_c3.txt
