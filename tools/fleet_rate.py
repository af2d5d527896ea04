"""Closed-loop rate of a FLEET (ltpl_fleet_*): N planners with device-resident state replay tick recordings of the reference from a
pre-uploaded tape, T ticks back to back without host synchronisation.   tools/fleet_rate.py [--planners 8192] [--ticks 200] [--mix]
Prints planner-ticks per second (device time between the first and the last launch) and checks planners of every group against the
recording's last tick."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                            # noqa: E402
import planner_replay as pr                                                   # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402


def group_inputs(lat, t):
    va = t['vel_args']
    return dict(prev_action=t['action_id_sel'], t_now=t['t'], vehicles=pr.vehicles_of_tick(t), zone_gids=pr.zone_gids_of_tick(lat, t),
                pos_est=t['pos_est'], vel_est=va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planners", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--mix", action="store_true", help="four groups: c2 / overtake / zonewall / c1 instead of c2 only")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--live", type=int, default=0, help="also time this many ticks through the per-call entry points (inputs handed over by the host every tick)")
    a = ap.parse_args()
    lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    hip = _capi.HipBackend(lat)
    names = ("c2", "overtake", "zonewall", "c1") if a.mix else ("c2",)
    recs = [pr.load_ticks(nm) for nm in names]
    n = a.planners
    sizes = [n // len(names)] * len(names)
    sizes[0] += n - sum(sizes)
    best = None
    for rep in range(a.reps):
        fleet = Fleet(hip, n)                     # (a fresh fleet per repetition: trajectory ids count up over a planner's life)
        t0 = time.perf_counter()
        for k in range(a.ticks):
            fleet.tape_append_groups([(sz, group_inputs(lat, ticks[k])) for sz, ticks in zip(sizes, recs)],
                                     ax_max_machines=recs[0][k]['vel_args']['ax_max_machines'])
        t_tape = time.perf_counter() - t0
        p = 0
        t0 = time.perf_counter()
        for sz, ticks in zip(sizes, recs):
            st = ticks[0]['start']
            for q in range(p, p + sz):
                fleet.set_start(q, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
            p += sz
        t_start = time.perf_counter() - t0
        t0 = time.perf_counter()
        ms = fleet.tape_run(0, a.ticks)
        wall = time.perf_counter() - t0
        best = ms if best is None else min(best, ms)
        print("rep %d: %d planners x %d ticks: device %.1f ms (wall %.1f ms) = %.3f M planner-ticks/s; %.3f ms per tick of the fleet "
              "(tape upload %.1f s, start poses %.1f s)" % (rep, n, a.ticks, ms, wall * 1e3, n * a.ticks / ms / 1e3, ms / a.ticks, t_tape, t_start))
        if rep + 1 < a.reps:
            fleet.close()
    p = 0
    for sz, ticks, nm in zip(sizes, recs, names):
        t = ticks[a.ticks - 1]
        for q in (p, p + sz - 1):
            traj, ids, ref = fleet.trajectories(q)
            pr.check_trajectories(traj, ids, ref, t, "%s tick %d planner %d" % (nm, t['tick'], q))
        p += sz
    print("parity: first and last planner of every group equal the recording's tick %d (trajectory digests%s)" % (
        a.ticks - 1, " + full arrays" if any(t['full'] is not None for t in [r[a.ticks - 1] for r in recs]) else ""))
    print("closed_loop_device_ticks_per_s %.0f" % (n * a.ticks / best * 1e3))
    if a.live:
        fleet.close()
        fleet = Fleet(hip, n)
        p = 0
        for sz, ticks in zip(sizes, recs):
            st = ticks[0]['start']
            for q in range(p, p + sz):
                fleet.set_start(q, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
            p += sz
        packed = [fleet.pack_groups([(sz, group_inputs(lat, ticks[k])) for sz, ticks in zip(sizes, recs)],
                                    ax_max_machines=recs[0][k]['vel_args']['ax_max_machines']) for k in range(a.live)]
        tp = tv = 0.0
        for k, (pi, vi, _keep) in enumerate(packed):
            t0 = time.perf_counter(); fleet.calc_paths_packed(pi); t1 = time.perf_counter(); fleet.calc_vel_profile_packed(vi); t2 = time.perf_counter()
            if k >= 5:
                tp += t1 - t0; tv += t2 - t1
        m = max(a.live - 5, 1)
        print("live inputs (per-call entry points, host wall time): calc_paths %.3f ms + calc_vel_profile %.3f ms per tick of the fleet = %.3f M "
              "planner-ticks/s" % (tp / m * 1e3, tv / m * 1e3, n * m / (tp + tv) / 1e6))
        traj, ids, ref = fleet.trajectories(0)
        pr.check_trajectories(traj, ids, ref, recs[0][a.live - 1], "live planner 0")


if __name__ == "__main__":
    main()
