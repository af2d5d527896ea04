"""Profiling helper (run under rocprofv3 on the GPU box): launches the seam-(1) kernel, the velocity kernel and the
fused tick kernel on the same C2 batch so that their per-kernel durations can be compared."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
scen, batch, vel = bench.make_batch(lat, n, seed=1)
res = None
for _ in range(reps):
    res = hip.plan_paths(batch, res)
hip.batch_upload(batch, vel)
hip.batch_run(reps=reps, timed=True)
# velocity seam alone: FB profiles over every produced path
jobs = []
for s in range(min(n, 2048)):
    for a in range(int(res.n_actions[s])):
        if res.valid[s, a]:
            npts = int(res.n_pts[s, a])
            pp = res.path_param[s, a, :npts]
            jobs.append({"mode": _capi.VEL_FB, "kappa": pp[:, 3].copy(), "el_lengths": pp[:-1, 4].copy(),
                         "loc_gg": np.ones((npts, 2)) * 5.0, "v_start": float(vel.vel_plan[s]), "v_end": 20.0})
for _ in range(reps):
    hip.vel_profile(vel.params, jobs)
print("profiled", n, "scenarios,", len(jobs), "fb jobs")
