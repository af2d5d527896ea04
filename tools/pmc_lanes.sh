#!/bin/bash
# PMC passes for the velocity kernels (wave cycles: waiting vs issuing; instruction counts), isolated (no overlap).
#   tools/pmc_lanes.sh <tag>
TAG=${1:-pl}
export TMPDIR=/tmp LTPL_NO_OVERLAP=1
OUT=$PWD/gpurun_out/${TAG}_pmc
mkdir -p $OUT
CMD="python bench.py --steps 6 --warmup 2 --no-cpu --latency-ticks 0 --dropin-ticks 0 --exact-steps --no-extra"
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o c -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/c_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if not k.startswith(("void k_", "k_")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if k in acc and r["Counter_Name"] == list(acc[k].keys())[0]: cnt[k] += 1
    for k, d in acc.items():
        print(k, "launches", cnt[k], {a: round(b / max(cnt[k], 1)) for a, b in d.items()})
PY
