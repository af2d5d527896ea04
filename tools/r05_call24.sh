#!/bin/bash
# round 5, closing GPU session on the final tree (the source of the r05x build): the default bench line with the committed counter files
# (build_matches), the two-rank check, the quick parity suite
export TMPDIR=/tmp
T=${R05TAG:-r05z}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_paths.py tests/test_gpu_configs.py tests/test_fresh_tick_golden.py tests/test_gpu_wave_ops.py -m gpu -x -q > gpurun_out/$T/gputest.txt 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/$T/gputest.txt
( time timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err ) 2> gpurun_out/$T/bench_time.txt; echo "bench rc=$?"; tail -3 gpurun_out/$T/bench_time.txt
python - <<PY
import json
d = json.loads(open("gpurun_out/$T/bench.json").readline())
r = d["roofline"]
print("value %.3f M  parity %s  frac %.3f  traffic %s traffic_build_matches %s" % (d["value"] / 1e6, d.get("parity_checked"), r["frac"], r["traffic"], r.get("traffic_build_matches")))
print("issue", {k: r["issue"].get(k) for k in ("valu_insts_per_scenario", "salu_insts_per_scenario", "lds_insts_per_scenario", "lanes_active", "valu_util", "lds_util", "wait_frac_of_wave_cycles", "build_matches")} if r.get("issue") else None)
print("binding", {k: v for k, v in r["binding"].items() if k != "what"})
e = d["extra"]
print("c3", {k: e["c3"].get(k) for k in ("ticks_per_s", "roofline_frac", "parity_checked", "traffic_frac", "traffic_build_matches")}, e["c3"].get("binding"))
print("fleet", e["closed_loop_device_mixed"]["planner_ticks_per_s"], e["closed_loop_device_mixed"]["planners_digested_on_device"], e["closed_loop_device_mixed"]["matches_recording"])
print("latency", {k: v for k, v in d["latency_us"].items() if k != "what"})
PY
