#!/bin/bash
# round 5, twenty-second GPU session: shell flush 16 x 8 (default) against 8 x 16 (variant ss16), parity of the mask; the driver-form bench line
export TMPDIR=/tmp
T=${R05TAG:-r05w}
mkdir -p gpurun_out/$T
V=graphbasedlocaltrajectoryplanner_amd/csrc/variants
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_edge_mask.py tests/test_fresh_tick_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/$T/gputest.txt
timeout 900 tools/ab_bench.sh base $V/r05i.so $V/ss16.so > gpurun_out/$T/ab_bench.txt 2>&1; cat gpurun_out/$T/ab_bench.txt
timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/%s/bench.json" % "${R05TAG:-r05w}").readline())
print("value %.2f M  ms/step %.4f" % (d["value"]/1e6, d["ms_per_step"]))
r=d["roofline"]; print("kernel_ms", r.get("kernel_ms"), "alone", r.get("kernel_ms_not_overlapped"), "frac", r.get("frac"), "build_matches", r.get("issue",{}).get("build_matches"))
e=d.get("extra",{})
for k in ("latency","dropin","c3","fleet","pcie"):
    if k in e: print(k, json.dumps(e[k])[:600])
PY
