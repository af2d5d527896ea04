#!/bin/bash
# Quick GPU-box check of a kernel change: (optionally) the GPU parity tests, then a short bench line and the per-kernel table.
#   tools/gpu_quick.sh <tag> [tests|notests] [bench args]
set -u
TAG=${1:-q}; T=${2:-tests}; shift 2 || true
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$T" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1
  echo "gpu tests rc=$?"; tail -4 gpurun_out/${TAG}_gputest.log
fi
ARGS=${@:-"--steps 100 --warmup 10 --no-cpu --latency-ticks 300 --dropin-ticks 600 --no-extra"}
timeout 600 python bench.py $ARGS > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("value %.3f M ticks/s  ms/step %.4f  frac %.3f" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"]))
print("pipeline", d["roofline"]["pipeline_ms"], "paths live", d["roofline"]["kernel_ms"])
print("latency", {k: (round(v,1) if isinstance(v,float) else v) for k,v in d["latency_us"].items() if k!="what"})
PY
tail -2 gpurun_out/${TAG}_bench.err
if [ "${LTPL_QUICK_STATS:-1}" = "1" ]; then
  # isolated per-kernel durations (no overlap between the path kernel and the velocity kernels)
  LTPL_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o k -- python bench.py --steps 30 --warmup 5 --no-cpu --latency-ticks 0 --exact-steps --no-extra > gpurun_out/${TAG}_stats.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/${TAG}_stats/**/k_kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if r["Name"].startswith(("void k_","k_")): print("%-44s calls %5s avg %10.1f us  min %9.1f" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
fi
