"""EXPERIMENT (round 4): does a fleet get faster when its planners are split over K independent fleets, each on its own handle = its own
HIP streams, their tapes running at the same time?   tools/fleet_streams.py [--planners 8192] [--ticks 200] [--fleets 1 2 4] [--mix]
A fleet's tick is a chain of ~10 latency-bound kernels of N short waves on one stream: consecutive kernels cannot overlap, and each ends
with a partially filled round. K chains next to each other fill those gaps -- the same sharding that puts planners on different GPUs
(DESIGN.md section 7), applied inside one GPU. Prints planner-ticks per second of host wall time around the K concurrent tape runs
(K = 1: also the device time of the one stream, for comparison with tools/fleet_rate.py) and checks the first / last planner of every
sub-fleet against the recording."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import planner_replay as pr                                                   # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402
from fleet_rate import group_inputs                                           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planners", type=int, default=8192)
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--fleets", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--mix", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    names = ("c2", "overtake", "zonewall", "c1") if a.mix else ("c2",)
    recs = [pr.load_ticks(nm) for nm in names]
    for K in a.fleets:
        n_sub = a.planners // K
        sizes = [n_sub // len(names)] * len(names)
        sizes[0] += n_sub - sum(sizes)
        hips = [_capi.HipBackend(lat) for _ in range(K)]
        best = None
        for rep in range(a.reps):
            fleets = [Fleet(h, n_sub) for h in hips]
            for fl in fleets:
                for k in range(a.ticks):
                    fl.tape_append_groups([(sz, group_inputs(lat, ticks[k])) for sz, ticks in zip(sizes, recs)],
                                          ax_max_machines=recs[0][k]['vel_args']['ax_max_machines'])
                p = 0
                for sz, ticks in zip(sizes, recs):
                    st = ticks[0]['start']
                    fl.set_start_range(p, p + sz, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
                    p += sz
            dev = [0.0] * K
            go = threading.Barrier(K + 1)

            def run(i):
                go.wait()
                dev[i] = fleets[i].tape_run(0, a.ticks)
            th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
            for t in th:
                t.start()
            go.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            wall = time.perf_counter() - t0
            rate = n_sub * K * a.ticks / wall
            best = rate if best is None else max(best, rate)
            print("K %d rep %d: %d x %d planners x %d ticks: wall %.1f ms = %.3f M planner-ticks/s (device time per stream: %s ms)" % (
                K, rep, K, n_sub, a.ticks, wall * 1e3, rate / 1e6, ", ".join("%.1f" % d for d in dev)))
            for fl in fleets:
                p = 0
                for sz, ticks, nm in zip(sizes, recs, names):
                    t = ticks[a.ticks - 1]
                    for q in (p, p + sz - 1):
                        traj, ids, ref = fl.trajectories(q)
                        pr.check_trajectories(traj, ids, ref, t, "%s tick %d planner %d" % (nm, t['tick'], q))
                    p += sz
                fl.close()
        print("fleets %d: best %.3f M planner-ticks/s (parity: first / last planner of every group of every sub-fleet equal the recording)" % (K, best / 1e6))
        for h in hips:
            h.close()


if __name__ == "__main__":
    main()
