#!/bin/bash
# round 5, fifth GPU session: generic profiles with direct output (k_vel_final for the follow tiles only): parity, same-box A/B, HBM bytes of the velocity kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/r05e
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05e/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r05e/gputest.txt
timeout 600 tools/ab_bench.sh $V/pre_emit.so base > gpurun_out/r05e/ab_bench.txt 2>&1; cat gpurun_out/r05e/ab_bench.txt
ARGS="--steps 20 --warmup 3 --no-cpu --latency-ticks 0 --dropin-ticks 0 --exact-steps --no-extra"
for C in FETCH_SIZE WRITE_SIZE; do
  for L in pre_emit base; do
    if [ $L = base ]; then unset LTPL_HIP_LIB; else export LTPL_HIP_LIB=$PWD/$V/$L.so; fi
    OUT=gpurun_out/r05e/pmc_${L}_$C; rm -rf $OUT; mkdir -p $OUT
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -o p -- python bench.py $ARGS > $OUT/run.log 2>&1
    python - "$OUT/p_counter_collection.csv" $L $C <<'PY' >> gpurun_out/r05e/vel_bytes.txt
import csv, sys, collections
acc = collections.defaultdict(list); grid = collections.defaultdict(int)
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid[k] = max(grid[k], int(r["Grid_Size"]))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if int(r["Grid_Size"]) == grid[k]: acc[k].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    if any(t in k for t in ("k_vel", "k_follow", "k_paths<1")): print("%-10s %-11s %-48s n=%3d mean %10.1f MB (KiB counter x 1024)" % (sys.argv[2], sys.argv[3], k[:48], len(v), sum(v) / len(v) * 1024 / 1e6))
PY
  done
done; unset LTPL_HIP_LIB; cat gpurun_out/r05e/vel_bytes.txt
