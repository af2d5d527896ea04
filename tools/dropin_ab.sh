#!/bin/bash
# A/B of the drop-in tick latency on ONE box: zero-copy output on / off
for Z in 0 1 0 1; do
  LTPL_ZC_OUT=$Z python - <<PY
import os, sys, numpy as np
sys.path.insert(0, ".")
import bench
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
lat = Lattice.load("tests/golden/monteblanco_lattice.npz")
hip = _capi.HipBackend(lat)
us, ok = bench.dropin_latency(hip, lat, 1200)
print("LTPL_ZC_OUT=%s dropin p50 %.1f p99 %.1f mean %.1f keys_ok %s" % (os.environ["LTPL_ZC_OUT"], np.percentile(us, 50), np.percentile(us, 99), us.mean(), ok))
PY
done
