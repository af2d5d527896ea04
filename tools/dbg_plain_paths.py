"""A few launches of the batch path kernel on the C2 (or, argument 2 = c3, the C3) bench batch (release library): target for profilers (tools/pc_sample.sh)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402

workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
if workload == "c3":                                                          # BASELINE config C3: the synthetic oval (bench.c3_throughput)
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice
    lat = c3_lattice()
else:
    lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
scen, batch, vel = bench.make_batch(lat, n, seed=1, workload=workload)
res = hip.new_paths_result(n)
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 10):
    hip.plan_paths(batch, res)
print("done", float(res.valid.sum()) / n)
