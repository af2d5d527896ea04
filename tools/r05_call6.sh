#!/bin/bash
# round 5, sixth GPU session: the default bench line of the build (all legs), k_vel_final row-chunk blocks A/B
export TMPDIR=/tmp
mkdir -p gpurun_out/r05f
( time timeout 900 python bench.py > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err ) 2> gpurun_out/r05f/bench_time.txt; echo "bench rc=$?"; tail -3 gpurun_out/r05f/bench_time.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05f/bench.json").readline())
print("value %.3f M  ms/step %.4f  parity %s  max_rel %.2e" % (d["value"] / 1e6, d["ms_per_step"], d.get("parity_checked"), d["parity_detail"]["max_rel_err"]))
print("elementwise", json.dumps(d["parity_detail"]["elementwise_rel_err"]))
print("latency", json.dumps({k: v for k, v in d["latency_us"].items() if k != "what"}))
e = d["extra"]
print("c3", {k: e["c3"][k] for k in ("ticks_per_s", "batch", "roofline_frac", "parity_checked")}, e["c3"].get("parity_detail", {}).get("elementwise_rel_err"))
print("fleet mixed", {k: e["closed_loop_device_mixed"][k] for k in ("planner_ticks_per_s", "planners", "matches_recording", "planner_checks", "planners_digested_on_device")}, e["closed_loop_device_mixed"].get("at_8192_planners"))
print("fleet same", {k: e["closed_loop_device"][k] for k in ("planner_ticks_per_s", "planners", "matches_recording", "live_inputs_planner_ticks_per_s")})
print("c5", {k: (v["p50_us"], v["p99_us"]) for k, v in e["c5"].items() if isinstance(v, dict)})
print("library", {k: d["library"].get(k) for k in ("path", "isa_sha256", "lib_sha256")}, "issue build_matches", (d["roofline"].get("issue") or {}).get("build_matches"))
PY
tail -5 gpurun_out/r05f/bench.err
ARGS="--steps 100 --warmup 10 --no-cpu --latency-ticks 0 --dropin-ticks 0 --no-extra"
for rep in 1 2; do for Y in 8 4 2 16; do LTPL_FINAL_Y=$Y python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('FINAL_Y=$Y  %.2f M ticks/s  k_paths live %.4f ms' % (d['value'] / 1e6, r['kernel_ms']))" >> gpurun_out/r05f/final_y.txt; done; done; cat gpurun_out/r05f/final_y.txt
