"""PCIe-inclusive rate of the host-buffer entry points on one MI355X: ltpl_tick_batch (capacity slabs) vs ltpl_tick_batch_compact
(trajectory rows in use, packed on the device, DMA into page-locked memory). Prints one JSON line."""
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
hip.tick_batch(batch, vel, res, vres)
t0 = time.perf_counter()
for _ in range(5):
    hip.tick_batch(batch, vel, res, vres)
slab = n * 5 / (time.perf_counter() - t0)
out = {"scenarios_per_call": n, "capacity_slab_ticks_per_s": slab}
for rows in (115, 0):
    comp = hip.new_compact_trajectories(n, max_rows=rows)
    hip.tick_batch_compact(batch, vel, comp)
    t0 = time.perf_counter()
    for _ in range(10):
        hip.tick_batch_compact(batch, vel, comp)
    out["compact_%s_ticks_per_s" % (rows or "all_rows")] = n * 10 / (time.perf_counter() - t0)
    out["compact_%s_bytes_per_tick" % (rows or "all_rows")] = float(comp.struct.total_rows) * 56 / n
print(json.dumps(out))
