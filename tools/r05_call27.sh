#!/bin/bash
# round 5, twenty-seventh GPU session: instruction-cache counters of the SINGLE TICK (k_tick, one 4-wave workgroup per launch) -- is a tick that
# runs ~12 k instructions per wave once, on an otherwise idle chip, bound by instruction fetch?
export TMPDIR=/tmp
T=${R05TAG:-r05D}
OUT=gpurun_out/$T/icache_tick; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu --latency-ticks 400 --dropin-ticks 0 --no-extra"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT -o p -- python bench.py $ARGS > $OUT/run.log 2>&1
python - "$OUT/p_counter_collection.csv" <<'PY' | tee gpurun_out/$T/icache_tick.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_tick" in r["Kernel_Name"]]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    by[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in by.items():
    n = len(next(iter(d.values())))
    print(k, "launches", n, " ".join("%s=%.0f" % (c.replace("SQC_", "").replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(d.items())), "(per launch = per tick, 4 waves)")
PY
