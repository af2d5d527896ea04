#!/bin/bash
# One rocprofv3 counter pass over a short bench run (kernel-trace only; no sys/hip/hsa trace domains).
#   tools/pmc_pass.sh <tag> "<COUNTER1 COUNTER2 ...>" [bench args]
set -u
TAG=$1; CTRS=$2; shift 2
ARGS=${@:-"--steps 20 --warmup 3 --no-cpu --latency-ticks 0 --exact-steps --no-extra"}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${TAG}
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o p -- python bench.py $ARGS > $OUT/run.log 2>&1
echo "rc=$?"
python - "$OUT/p_counter_collection.csv" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "copyBuffer" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%4d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
PY
