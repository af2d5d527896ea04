#!/bin/bash
# HBM bytes per launch of every kernel of the tick pipeline (FETCH_SIZE, WRITE_SIZE in their own passes, MI355X_MICROARCH.md), per library variant:
#   tools/vel_bytes.sh <lib|base> ...      -> per kernel: fetch MB (counter KiB x 1024 x 2: the gfx950 correction of FETCH_SIZE), write MB, sum
export TMPDIR=/tmp LTPL_NO_OVERLAP=1
CMD="python bench.py --steps 6 --warmup 2 --no-cpu --latency-ticks 0 --dropin-ticks 0 --exact-steps --no-extra"
for V in "$@"; do
  if [ "$V" = base ]; then unset LTPL_HIP_LIB; T=base; else export LTPL_HIP_LIB=$PWD/$V; T=$(basename $V .so); fi
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=gpurun_out/velbytes_${T}_$C; rm -rf $OUT; mkdir -p $OUT
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o c -- $CMD > $OUT/run.log 2>&1
  done
  python - "$T" <<'PY'
import csv, glob, collections, sys
T = sys.argv[1]
acc = {c: collections.defaultdict(list) for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob("gpurun_out/velbytes_%s_%s/**/c_counter_collection.csv" % (T, c), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith(("void k_", "k_")) and r["Counter_Name"] == c and int(r.get("Grid_Size", "0") or 0) > 4096:
                acc[c][k[:44]].append(float(r["Counter_Value"]))
tot = 0.0
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    fe = acc["FETCH_SIZE"].get(k, [0.0]); wr = acc["WRITE_SIZE"].get(k, [0.0])
    f_mb = 2.0 * 1024.0 * sum(fe) / len(fe) / 1e6; w_mb = 1024.0 * sum(wr) / len(wr) / 1e6
    if "k_paths" not in k: tot += f_mb + w_mb
    print("%-8s %-46s fetch %8.1f MB  write %8.1f MB  sum %8.1f MB  (n = %d)" % (T, k, f_mb, w_mb, f_mb + w_mb, len(fe)))
print("%-8s velocity stage (everything but k_paths): %.1f MB per step" % (T, tot))
PY
done
