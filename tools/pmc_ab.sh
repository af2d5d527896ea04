#!/bin/bash
# Counter passes over a few launches of the batch path kernel for several library variants on ONE box:
#   tools/pmc_ab.sh <lib|base> ...      -> per scenario: VALU / SALU / LDS / SMEM instructions, active lanes, wave / wait cycles
# (two passes per variant: the SQ block holds ~8 counters at a time)
export TMPDIR=/tmp
N=${PMC_N:-32768}
for V in "$@"; do
  if [ "$V" = base ]; then unset LTPL_HIP_LIB; T=base; else export LTPL_HIP_LIB=$PWD/$V; T=$(basename $V .so); fi
  for PASS in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"; do
    OUT=gpurun_out/pmcab_${T}_$(echo $PASS | cut -c4-12); rm -rf $OUT; mkdir -p $OUT
    timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT -o p -- python tools/dbg_plain_paths.py $N c2 4 > $OUT/run.log 2>&1
    python - "$OUT/p_counter_collection.csv" "$T" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_paths<1" in r["Kernel_Name"]]
acc = collections.defaultdict(list)
for r in rows:
    if int(r["Grid_Size"]) >= 64 * 1024:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]) / (int(r["Grid_Size"]) / 64))
print(sys.argv[2], " ".join("%s=%.0f" % (c.replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(acc.items())))
# per-LAUNCH means of the full-batch launches -> gpurun_out/pmc_issue_<variant>.json (what bench.py reads as profiles/pmc_issue.json)
import json, os
grid = max(int(r["Grid_Size"]) for r in rows) if rows else 0
per = {c: sum(v) / len(v) * (grid / 64) for c, v in acc.items()}
dst = os.path.join("gpurun_out", "pmc_issue_%s.json" % sys.argv[2])
d = json.load(open(dst)) if os.path.isfile(dst) else {"kernel": "k_paths<1>", "tag": os.environ.get("PMC_TAG", "r04"), "workload": "c2", "grid_size": grid,
                                                      "what": "rocprofv3 --pmc passes over tools/dbg_plain_paths.py (path kernel alone), mean per launch; "
                                                              "SQ_*_CYCLES / SQ_ACTIVE_* in units of 4 cycles (quad-cycles)"}
m = {"SQ_INSTS_VALU": "valu_insts_per_launch", "SQ_INSTS_SALU": "salu_insts_per_launch", "SQ_INSTS_LDS": "lds_insts_per_launch",
     "SQ_INSTS_SMEM": "smem_insts_per_launch", "SQ_ACTIVE_INST_VALU": "valu_active_quad_cycles_per_launch",
     "SQ_ACTIVE_INST_LDS": "lds_active_quad_cycles_per_launch", "SQ_ACTIVE_INST_SCA": "scalar_active_quad_cycles_per_launch",
     "SQ_WAVE_CYCLES": "wave_quad_cycles_per_launch", "SQ_WAIT_ANY": "wait_any_quad_cycles_per_launch",
     "SQ_WAIT_INST_ANY": "wait_inst_quad_cycles_per_launch", "SQ_LDS_BANK_CONFLICT": "lds_bank_conflict_cycles_per_launch",
     "SQ_BUSY_CYCLES": "sq_busy_quad_cycles_per_launch"}
for c, k in m.items():
    if c in per:
        d[k] = per[c]
if "SQ_THREAD_CYCLES_VALU" in per and per.get("SQ_ACTIVE_INST_VALU"):
    d["lanes_active_per_valu_inst"] = per["SQ_THREAD_CYCLES_VALU"] / per["SQ_ACTIVE_INST_VALU"]
json.dump(d, open(dst, "w"), indent=1)
PY
  done
done
