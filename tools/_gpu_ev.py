import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
scen, batch, vel = bench.make_batch(lat, 32768, seed=1, workload="c2")
hip.batch_upload(batch, vel)
hip.batch_run(reps=30, timed=False)
for rep in range(2):
    ms = hip.batch_run(reps=300, timed=True) / 300
    print("LTPL_EXP_PIPE=%s: %.4f ms/step, k_paths live %.4f" % (os.environ.get("LTPL_EXP_PIPE", "0"), ms, hip.batch_last_paths_ms()))
