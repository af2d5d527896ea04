"""Closed-loop rate of a BATCH of planners (ltpl_planner_* entry points): N planners are driven with the recorded inputs of the C2 loop
(every planner the same inputs; each carries its own state), wall time per tick and planner-ticks per second (the
fleet, tools/fleet_rate.py, is the form meant for batches):

    python tools/planner_batch_rate.py --planners 256             # MI355X
    python tools/planner_batch_rate.py --planners 64 --harness    # no GPU (oracle arithmetic)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planners", type=int, default=256)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--harness", action="store_true", help="host-logic harness with the oracle's arithmetic instead of the GPU (test infrastructure)")
    a = ap.parse_args()
    import planner_replay as pr
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    ticks = pr.load_ticks("c2")[:a.ticks]
    n = a.planners
    if a.harness:
        from oracle.planner_host import HostPlannerBackend
        pl = HostPlannerBackend(lat).planner(n)
    else:
        from graphbasedlocaltrajectoryplanner_amd import _capi
        from graphbasedlocaltrajectoryplanner_amd.planner import Planner
        pl = Planner(_capi.HipBackend(lat), n)
    st = ticks[0]['start']
    for s in range(n):
        pl.set_start(s, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    t_c, t_v = [], []
    for t in ticks:
        veh = [pr.vehicles_of_tick(t)] * n
        zg = [pr.zone_gids_of_tick(lat, t)] * n
        va = t['vel_args']
        t0 = time.perf_counter()
        pl.calc_paths([t['action_id_sel']] * n, [t['t']] * n, veh, zg)
        t1 = time.perf_counter()
        pl.calc_vel_profile([t['pos_est']] * n, va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                            ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])
        t2 = time.perf_counter()
        t_c.append(t1 - t0); t_v.append(t2 - t1)
    a0, a1 = pl.trajectories(0), pl.trajectories(n - 1)
    same = list(a0[0].keys()) == list(a1[0].keys()) and all(np.array_equal(a0[0][k][0], a1[0][k][0]) for k in a0[0])
    c, v = np.array(t_c[20:]), np.array(t_v[20:])
    print("planners %d  calc_paths %.3f ms  calc_vel_profile %.3f ms  tick %.3f ms  = %.0f planner-ticks/s  (Python packing of "
          "the inputs included; first / last planner identical: %s)" % (n, c.mean() * 1e3,
                                                                        v.mean() * 1e3, (c + v).mean() * 1e3, n / (c + v).mean(), same))


if __name__ == "__main__":
    main()
