import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
lat = Lattice.load("/root/repo/tests/golden/monteblanco_lattice.npz")
hip = _capi.HipBackend(lat)
scen, batch, vel = bench.make_batch(lat, 2048, seed=1)
res = hip.plan_paths(batch, hip.new_paths_result(2048))
aid = np.asarray(res.action_id).reshape(2048, -1); valid = np.asarray(res.valid).reshape(2048, -1); red = np.asarray(res.reduced).reshape(2048, -1)
fol = (aid == _capi.ACT_FOLLOW) & (valid != 0)
print("scenarios with a follow path: %.3f; of those reduced: %.3f; valid paths per scenario %.2f; any reduced path %.3f" % (
    fol.any(1).mean(), (fol & (red != 0)).any(1).sum() / max(fol.any(1).sum(), 1), (valid != 0).sum(1).mean(), ((valid != 0) & (red != 0)).any(1).mean()))
