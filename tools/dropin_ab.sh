#!/bin/bash
# A/B of the drop-in tick latency on ONE box: every argument is one environment setting ("LTPL_ZC_OUT=0", "A=1 B=2", "" = defaults);
# the settings are run alternately, twice each.   tools/dropin_ab.sh "" "LTPL_NW1_MIN_SCEN=1"
[ $# -eq 0 ] && set -- "LTPL_ZC_OUT=0" "LTPL_ZC_OUT=1"
for rep in 1 2; do
for S in "$@"; do
  env $S LTPL_AB_TAG="$S" python - <<PY
import os, sys, numpy as np
sys.path.insert(0, ".")
import bench
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
lat = Lattice.load("tests/golden/monteblanco_lattice.npz")
hip = _capi.HipBackend(lat)
us, ok = bench.dropin_latency(hip, lat, 1200)
print("[%s] dropin p50 %.1f p99 %.1f mean %.1f keys_ok %s" % (os.environ["LTPL_AB_TAG"], np.percentile(us, 50), np.percentile(us, 99), us.mean(), ok))
PY
done
done
