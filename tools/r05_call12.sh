#!/bin/bash
# round 5, twelfth GPU session: what binds the fleet's stage kernels -- SQ counters per kernel over a short mixed tape of 32 768 planners
export TMPDIR=/tmp
mkdir -p gpurun_out/r05l
CMD="python tools/fleet_rate.py --planners 32768 --ticks 24 --mix --reps 1"
for PASS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  OUT=gpurun_out/r05l/$(echo $PASS | tr ' ' '_' | cut -c1-20); rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT -o p -- $CMD > $OUT/run.log 2>&1
  python - "$OUT/p_counter_collection.csv" <<'PY' >> gpurun_out/r05l/fleet_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); grid = collections.defaultdict(int)
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", ""); grid[k] = max(grid[k], int(r["Grid_Size"]))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if int(r["Grid_Size"]) == grid[k]: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "rocclr" in k or "k_paths<4" in k: continue
    n = grid[k] / 64
    print("%-44s waves %7d  " % (k[:44], n) + "  ".join("%s=%.0f" % (c.replace("SQ_", ""), sum(v) / len(v) / n) for c, v in sorted(d.items())) + "  (per wave)")
PY
done; cat gpurun_out/r05l/fleet_pmc.txt
