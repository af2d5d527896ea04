#!/bin/bash
# Counter passes over a short fleet run (run on the GPU box through gpurun): HBM bytes and instruction counts per kernel of the fleet tick.
#   tools/fleet_pmc.sh <tag>      -> gpurun_out/fleet_pmc_<tag>/{fetch,write,insts}/..., summary printed and written as <tag>_fleet_pmc.json
set -u
TAG=${1:-r03g}
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/fleet_pmc_$TAG
mkdir -p $OUT
CMD="python $ROOT/tools/fleet_rate.py --planners 8192 --ticks 30 --reps 1"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/insts -o p -- $CMD > $OUT/insts.log 2>&1; echo "insts rc=$?"
cd $ROOT
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for sub in ("fetch", "write", "insts"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(acc.items()):
    if "rocclr" in k:
        continue
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    # FETCH_SIZE / WRITE_SIZE: KB per launch on this stack (MI355X_MICROARCH.md, HBM section: 64-byte units reported in KB; gfx950 correction
    # of the fetch counter as tools/profile_summarise.py applies it)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_mb_per_launch_raw"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024 / 1e6
        e["hbm_mb_per_launch_corrected"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024 / 1e6     # (x2 on the fetch side: upper bound, as profile_summarise.py)
    res[k] = e
    print("%-44s %s" % (k[:44], {c: round(v, 1) for c, v in e.items()}))
json.dump({"tag": tag, "command": "tools/fleet_rate.py --planners 8192 --ticks 30", "per_launch": res}, open("gpurun_out/%s_fleet_pmc.json" % tag, "w"), indent=1)
PY
