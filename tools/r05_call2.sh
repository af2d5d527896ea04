#!/bin/bash
# round 5, second GPU session: phase stamps of the path kernel on C2 and C3, instruction-cache counters, fleet tests after the seen_gg fix
export TMPDIR=/tmp
mkdir -p gpurun_out/r05b
timeout 300 python tools/dbg_paths_timing.py 32768 c2 > gpurun_out/r05b/phases_c2.txt 2>&1; grep "ltpl dbg" gpurun_out/r05b/phases_c2.txt | tail -6
timeout 300 python tools/dbg_paths_timing.py 8192 c3 > gpurun_out/r05b/phases_c3.txt 2>&1; grep "ltpl dbg" gpurun_out/r05b/phases_c3.txt | tail -6
OUT=gpurun_out/r05b/icache; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAVE_CYCLES --output-format csv -d $OUT -o p -- python tools/dbg_plain_paths.py 32768 c2 4 > $OUT/run.log 2>&1
python tools/pmc_summarise.py $OUT/p_counter_collection.csv icache c2 graphbasedlocaltrajectoryplanner_amd/csrc/libltpl_hip.so > gpurun_out/r05b/icache.txt 2>&1; cat gpurun_out/r05b/icache.txt
timeout 900 python -m pytest tests/test_fleet_differential.py tests/test_gpu_fleet.py -m gpu -x -q > gpurun_out/r05b/fleettest.txt 2>&1; echo "fleet tests rc=$?"; tail -4 gpurun_out/r05b/fleettest.txt
