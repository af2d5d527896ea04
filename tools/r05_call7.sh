#!/bin/bash
# round 5, final GPU session of the build: parity suite, the default bench line, kernel stats, counter + traffic passes (C2, C3) stamped with
# the build, fleet kernel stats. Outputs: gpurun_out/r05g/*, gpurun_out/pmc_{issue,traffic}_base[_c3].json
export TMPDIR=/tmp
T=${1:-r05g}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/$T/gputest.txt
( time timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err ) 2> gpurun_out/$T/bench_time.txt; echo "bench rc=$?"; head -c 600 gpurun_out/$T/bench.json; echo; tail -3 gpurun_out/$T/bench_time.txt
ARGS="--steps 50 --warmup 5 --no-cpu --latency-ticks 200 --dropin-ticks 300 --exact-steps --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/stats -o k -- python bench.py $ARGS > gpurun_out/$T/stats.log 2>&1; echo "stats rc=$?"
find gpurun_out/$T/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$T/kernel_stats.csv; head -12 gpurun_out/$T/kernel_stats.csv | cut -c1-160
PMC_TAG=$T timeout 900 tools/pmc_ab.sh base > gpurun_out/$T/pmc_c2.txt 2>&1; cat gpurun_out/$T/pmc_c2.txt
PMC_TAG=$T PMC_WORKLOAD=c3 PMC_N=32768 timeout 900 tools/pmc_ab.sh base > gpurun_out/$T/pmc_c3.txt 2>&1; cat gpurun_out/$T/pmc_c3.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/fleet_stats -o k -- python tools/fleet_rate.py --planners 32768 --ticks 50 --mix --reps 1 > gpurun_out/$T/fleet_stats.log 2>&1; echo "fleet stats rc=$?"
find gpurun_out/$T/fleet_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$T/fleet_kernel_stats.csv; head -16 gpurun_out/$T/fleet_kernel_stats.csv | cut -c1-150
python tools/c5_latency.py 300 300 2>/dev/null | tail -2
