"""Where one closed-loop tick of the planner entry points spends its time (host wall clock per step, MI355X)."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
from graphbasedlocaltrajectoryplanner_amd.planner import Planner
from oracle.fixture_io import load_records
import planner_replay as pr
lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
ticks = load_records(os.path.join(ROOT, "tests", "golden", "c2_ticks.npz"))[:1200]
pl = Planner(hip, 1)
st = ticks[0]['start']; pl.set_start(0, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
T = {k: [] for k in ("pack", "c_paths", "paths()", "pack_vel+c_vel", "traj()")}
for t in ticks:
    veh = pr.vehicles_of_tick(t); zg = pr.zone_gids_of_tick(lat, t); va = t['vel_args']
    t0 = time.perf_counter(); i, keep = pl._pack_paths_in([t['action_id_sel']], [t['t']], [veh], [zg]); t1 = time.perf_counter()
    pl._check(pl._fn("calc_paths")(pl.handle, C.byref(i))); t2 = time.perf_counter()
    pl.paths(0); t3 = time.perf_counter()
    pl.calc_vel_profile([t['pos_est']], va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                        ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d']); t4 = time.perf_counter()
    pl.trajectories(0); t5 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        T[k].append(v * 1e6)
print({k: "p50 %.1f p99 %.1f" % (np.percentile(v[100:], 50), np.percentile(v[100:], 99)) for k, v in T.items()})
