#!/bin/bash
# tools/insts_per_layer.sh [n_scen] [no_opponents]  -> per-dispatch counters of k_paths in launch order (3 launches per H)
export TMPDIR=/tmp
OUT=gpurun_out/ipl; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT -o p -- python tools/insts_per_layer.py "$@" > $OUT/run.log 2>&1
grep "^H" $OUT/run.log
python - "$OUT/p_counter_collection.csv" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_paths<1" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    by[r["Dispatch_Id"]]["grid"] = int(r["Grid_Size"])
for k, d in by.items():
    n = d["grid"] / 64
    print(k, " ".join("%s=%.0f" % (c.replace("SQ_", ""), v / n) for c, v in sorted(d.items()) if c != "grid"))
PY
