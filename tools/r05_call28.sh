#!/bin/bash
# round 5, twenty-eighth GPU session: does the single tick keep its instruction cache without the copy in front of it? zero-copy inputs
# (LTPL_ZC_IN=1) against the default, latency A/B + the instruction fetch counter
export TMPDIR=/tmp
T=${R05TAG:-r05E}
mkdir -p gpurun_out/$T
timeout 400 tools/tick_ab.sh "" "LTPL_ZC_IN=1" > gpurun_out/$T/tick_zcin_ab.txt 2>&1; cat gpurun_out/$T/tick_zcin_ab.txt
OUT=gpurun_out/$T/icache_zcin; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu --latency-ticks 400 --dropin-ticks 0 --no-extra"
LTPL_ZC_IN=1 timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAVE_CYCLES --output-format csv -d $OUT -o p -- python bench.py $ARGS > $OUT/run.log 2>&1
python - "$OUT/p_counter_collection.csv" <<'PY' | tee gpurun_out/$T/icache_zcin.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_tick" in r["Kernel_Name"]]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    by[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in by.items():
    n = len(next(iter(d.values())))
    print("LTPL_ZC_IN=1", k, "launches", n, " ".join("%s=%.0f" % (c.replace("SQC_", "").replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(d.items())), "(per tick)")
PY
