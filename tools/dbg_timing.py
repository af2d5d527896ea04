"""Run small batches with LTPL_DEBUG_TIMING=1 to print in-kernel phase cycle counts (see dbg_report in ltpl_hip.hip)."""
import os
import sys
import numpy as np
os.environ["LTPL_DEBUG_TIMING"] = "1"
# the switch only exists in the experiment build of the library (include/ltpl_hip.h)
os.environ.setdefault("LTPL_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                   "graphbasedlocaltrajectoryplanner_amd", "csrc", "libltpl_hip_exp.so"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
scen, batch, vel = bench.make_batch(lat, n, seed=1)
for _ in range(2):
    res = hip.plan_paths(batch)
for _ in range(2):
    res, vres = hip.tick_batch(batch, vel)
jobs = []
for s in range(n):
    for a in range(int(res.n_actions[s])):
        if res.valid[s, a]:
            npts = int(res.n_pts[s, a])
            pp = res.path_param[s, a, :npts]
            jobs.append({"mode": _capi.VEL_FB, "kappa": pp[:, 3].copy(), "el_lengths": pp[:-1, 4].copy(),
                         "loc_gg": np.ones((npts, 2)) * 5.0, "v_start": float(vel.vel_plan[s]), "v_end": 20.0})
jobs = jobs[:256]
for _ in range(2):
    hip.vel_profile(vel.params, jobs)
print("n_pts mean", float(res.n_pts[res.valid == 1].mean()), "jobs", len(jobs))
