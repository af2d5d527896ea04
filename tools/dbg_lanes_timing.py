"""In-kernel phase cycle counts of the batch velocity kernel (k_vel_lanes), LTPL_DEBUG_TIMING=1 (100 MHz clock64 ticks)."""
import os
import sys
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
scen, batch, vel = bench.make_batch(lat, n, seed=1)
res, vres = hip.new_paths_result(n), _capi.TickVelResult(n, hip.caps.max_path_pts)
for i in range(3):
    hip.tick_batch(batch, vel, res, vres)
