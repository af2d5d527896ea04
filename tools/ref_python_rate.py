#!/usr/bin/env python
"""Reference-Python CPU baseline as a committed measurement ("reference-Python-over-shim").

BUILD CONTAINER ONLY (needs /root/reference; never runs on the GPU box and is not reachable from bench.py's timed region).
The UNMODIFIED reference (graph_ltpl imported from /root/reference through oracle/ref_env.py, on the restated igraph /
trajectory_planning_helpers shims -- the real wheels are not installed here, so the label is "over-shim", not "reference") runs the C2
closed loop of main_std_example.py: Monteblanco, sample zone, 8 dynamic opponents, dt = 50 ms (oracle/ref_scenarios.run_loop, the same
loop the golden fixtures were recorded from). `time.perf_counter` is read around `Graph_LTPL.calc_paths` and `Graph_LTPL.calc_vel_profile`
ONLY (the hot path of BASELINE.json's north_star: graph search + spline + velocity profile; the opponent simulators and the vehicle
dummy of the loop are outside the timers). >= 2 000 ticks after 100 warm-up ticks, one core.

    python tools/ref_python_rate.py [--ticks 2000] [--warmup 100] [--out profiles/r05_ref_python_cpu.json]
"""
import argparse
import json
import os
import platform
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_ref_python_cpu.json"))
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # one core: BLAS threads off (set before numpy work starts)
        os.environ.setdefault(k, "1")
    from oracle import ref_env, ref_scenarios as rs
    if not ref_env.reference_available():
        raise SystemExit("tools/ref_python_rate.py: /root/reference is not present (build container only)")
    cache = os.path.join(ROOT, "oracle", "_cache")
    gl, clock, ltpl_obj, gb, path_dict = rs.make_planner(cache)

    t_paths, t_vel = [], []
    cp, cv = ltpl_obj.calc_paths, ltpl_obj.calc_vel_profile

    def timed_paths(*a, **k):
        t0 = time.perf_counter()
        r = cp(*a, **k)
        t_paths.append(time.perf_counter() - t0)
        return r

    def timed_vel(*a, **k):
        t0 = time.perf_counter()
        r = cv(*a, **k)
        t_vel.append(time.perf_counter() - t0)
        return r

    ltpl_obj.calc_paths, ltpl_obj.calc_vel_profile = timed_paths, timed_vel          # (instance attributes: the class is untouched)
    n = args.warmup + args.ticks
    w0 = time.perf_counter()
    exported = rs.run_loop(gl, clock, ltpl_obj, path_dict, n_ticks=n, dt=0.05, dummies=rs.opponents_c2(gl, 8), zones=rs.ZONE_EXAMPLE)
    wall = time.perf_counter() - w0
    assert len(t_paths) == n and len(t_vel) == n and len(exported) == n
    tick = (np.array(t_paths) + np.array(t_vel))[args.warmup:]
    tp, tv = np.array(t_paths)[args.warmup:], np.array(t_vel)[args.warmup:]
    import collections
    sets = collections.Counter(tuple(sorted(e['traj'].keys())) for e in exported[args.warmup:])

    def stats(x):
        return {"mean_ms": float(x.mean() * 1e3), "p50_ms": float(np.percentile(x, 50) * 1e3), "p99_ms": float(np.percentile(x, 99) * 1e3),
                "max_ms": float(x.max() * 1e3)}

    out = {"what": "UNMODIFIED reference graph_ltpl (imported from /root/reference) over the restated igraph / trajectory_planning_helpers shims "
                   "of oracle/shims (the real wheels are not installed: 'reference-Python-over-shim', BASELINE.md section 4), C2 closed loop "
                   "(Monteblanco, sample zone, 8 dynamic opponents, dt = 50 ms; oracle/ref_scenarios.run_loop); time.perf_counter around "
                   "Graph_LTPL.calc_paths + Graph_LTPL.calc_vel_profile only",
           "ticks": int(args.ticks), "warmup_ticks": int(args.warmup), "cores": 1,
           "cpu_model": cpu_model(), "python": platform.python_version(), "numpy": np.__version__,
           "ticks_per_s": float(1.0 / tick.mean()), "tick": stats(tick), "calc_paths": stats(tp), "calc_vel_profile": stats(tv),
           "whole_loop_ticks_per_s_incl_simulators": float(n / wall),
           "offered_action_sets": {" + ".join(k): int(v) for k, v in sorted(sets.items(), key=lambda kv: -kv[1])},
           "script": "tools/ref_python_rate.py", "where": "build container (no GPU); not the GPU box's host"}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
