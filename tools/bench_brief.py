#!/usr/bin/env python
"""One-screen summary of a bench.py JSON line (tools/gpu_session.sh): headline, path kernel, parity, latency, the extra legs."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print("value %.3f M %s  ms/step %.4f  n_gpus %d  scaling %s  dtype %s" % (d["value"] / 1e6, d["unit"], d["ms_per_step"], d["n_gpus"], d["scaling"], d["dtype"]))
print("k_paths live %.4f ms  alone %.4f ms  frac %.3f  bound %s  pipeline %s" % (r["kernel_ms"], r["kernel_ms_not_overlapped"], r["frac"], r["bound"],
                                                                              {k: round(v, 4) for k, v in r["pipeline_ms"].items()}))
if "parity_detail" in d:
    e = d["parity_detail"]["elementwise_rel_err"]
    print("parity %s  max_rel %.2e  elementwise max: vx %.2e  ax %.2e  kappa %.2e  int mismatches %s" % (
        d["parity_checked"], d["parity_detail"]["max_rel_err"], e["vx"]["max"] or 0, e["ax"]["max"] or 0, e["kappa"]["max"] or 0, d["parity_detail"]["integer_mismatches"]))
if "cpu_baseline" in d:
    print("cpu_baseline %.0f ticks/s (%s, %d core)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"]))
print("latency_us", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["latency_us"].items() if k != "what"})
x = d.get("extra", {})
for k in ("three_slot_ticks_per_s",):
    if k in x:
        print(k, "%.2f M" % (x[k] / 1e6))
if "pcie_inclusive" in x:
    print("pcie_inclusive %.2f M ticks/s" % (x["pcie_inclusive"]["ticks_per_s"] / 1e6))
for k in ("closed_loop_device", "closed_loop_device_mixed"):
    if k in x:
        print(k, "%.2f M planner-ticks/s" % (x[k]["planner_ticks_per_s"] / 1e6), "ok", x[k]["matches_recording"],
              ("live %.2f M" % (x[k]["live_inputs_planner_ticks_per_s"] / 1e6)) if x[k].get("live_inputs_planner_ticks_per_s") else "",
              ("at 8192: %.2f M" % (x[k]["at_8192_planners"]["planner_ticks_per_s"] / 1e6)) if "at_8192_planners" in x[k] else "")
if "c4" in x:
    for k in ("one_gpu", "shard"):
        c = x["c4"][k]
        print("c4 %-8s n %5d  resident %.2f M ticks/s (%.1f us/step)  pcie %.2f M ticks/s (p50 %.1f us, p99 %.1f us per call)" % (
            k, c["scenarios"], c["resident_ticks_per_s"] / 1e6, c["resident_us_per_step"], c["pcie_ticks_per_s"] / 1e6, c["pcie_us_per_call_p50"], c["pcie_us_per_call_p99"]))
if "c3" in x:
    c = x["c3"]
    e = (c.get("parity_detail") or {}).get("elementwise_rel_err", {})
    print("c3 %.2f M ticks/s  kernel %.3f ms  frac %.3f  parity %s  vx %.2e ax %.2e" % (c["ticks_per_s"] / 1e6, c["kernel_ms"], c["roofline_frac"], c.get("parity_checked"),
                                                                                     (e.get("vx") or {}).get("max") or 0, (e.get("ax") or {}).get("max") or 0))
if "c5" in x:
    print("c5 300m p50 %.0f us p99 %.0f us; 100m p50 %.0f us" % (x["c5"]["horizon_300m"]["p50_us"], x["c5"]["horizon_300m"]["p99_us"], x["c5"]["horizon_100m"]["p50_us"]))
if "persistent_tick" in d["latency_us"]:
    print("persistent_tick", d["latency_us"]["persistent_tick"])
