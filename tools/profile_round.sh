#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel stats + HBM traffic counters for the bench command.
#   tools/profile_round.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,fetch,write}/... ; tools/profile_summarise.py turns them into profiles/<tag>_*.
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa trace domains).
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:-"--steps 50 --warmup 5 --no-cpu --latency-ticks 200 --dropin-ticks 300 --exact-steps --no-extra"}
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python bench.py $ARGS > $OUT/stats.log 2>&1
echo "stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python bench.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python bench.py $ARGS > $OUT/write.log 2>&1
echo "write rc=$?"
find $OUT -name "*.csv" | head -20
