"""In-kernel phase cycle counts of the path kernel (LTPL_DEBUG_TIMING=1, shader-clock ticks, blocks 0..255 of the launch).
Stamps: 0 start | 1 scenario set-up | 2 closest layer per position | 3 edge mask | 4 closest object / template |
5 sweeps | 6 goal / horizon decisions | 7 path assembly.   tools/dbg_paths_timing.py [n_scen] [workload]"""
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

workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
if workload == "c3":
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice
    lat = c3_lattice()
else:
    lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
scen, batch, vel = bench.make_batch(lat, n, seed=1, workload=workload)
res = hip.new_paths_result(n)
for i in range(2):
    hip.plan_paths(batch, res)
