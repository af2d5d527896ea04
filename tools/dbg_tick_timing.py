"""In-kernel phase cycle counts of the single-scenario latency kernel (k_tick), LTPL_DEBUG_TIMING=1."""
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

lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
scen, batch, vel = bench.make_batch(lat, 64, seed=1)
res, vres = hip.new_paths_result(1), _capi.TickVelResult(1, hip.caps.max_path_pts)
for i in range(24):
    b1 = _capi.PathsBatch([scen[i]], w_last_edges=[0.0, 0.5, 0.8])
    v1 = _capi.TickVelBatch(vel.params, 1, vel.vel_plan[i:i + 1], vel.vel_est[i:i + 1],
                            np.array([[vel.pos_x[i], vel.pos_y[i]]]), vel.veh_vel[batch.veh_off[i]:batch.veh_off[i + 1]])
    hip.tick_batch(b1, v1, res, vres)
