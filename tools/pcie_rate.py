"""PCIe-inclusive throughput of the host-buffer entry point: ltpl_tick_batch on a batch of C2 scenarios, host wall time per
call including packing, H2D, kernels, D2H and unpacking (never the bench `value`)."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
scen, batch, vel = bench.make_batch(lat, n, seed=1)
res, vres = hip.new_paths_result(n), _capi.TickVelResult(n, hip.caps.max_path_pts)
for _ in range(3):
    hip.tick_batch(batch, vel, res, vres)
t0 = time.perf_counter(); reps = 10
for _ in range(reps):
    hip.tick_batch(batch, vel, res, vres)
el = (time.perf_counter() - t0) / reps
out_bytes = sum(a.nbytes for a in (res.nodes, res.node_idx, res.coeff, res.path_param, vres.vx, vres.ax))
print(json.dumps({"batch": n, "ms_per_call": el * 1e3, "ticks_per_s_pcie_inclusive": n / el,
                  "output_bytes_per_call": out_bytes, "output_GBps": out_bytes / el / 1e9}))
