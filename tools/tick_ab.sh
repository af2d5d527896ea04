#!/bin/bash
# A/B of the single-tick latency (bench.single_tick_latency: one synchronous ltpl_tick_batch call incl. packing / unpacking) and of the
# drop-in tick on ONE box: every argument is one environment setting ("" = defaults, "LTPL_TICK_GRAPH=1", "LTPL_POLL=1 LTPL_TICK_GRAPH=1");
# the settings are run alternately, twice each.   tools/tick_ab.sh "" "LTPL_TICK_GRAPH=1"      ("LTPL_PERSISTENT_TICK=1": the resident kernel, round 6)
[ $# -eq 0 ] && set -- "" "LTPL_TICK_GRAPH=1"
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
scen, batch, vel = bench.make_batch(lat, 64, seed=7)
us, single = bench.single_tick_latency(hip, lat, scen, vel, batch, 2000)
hip.batch_upload(single[0], single[1]); hip.batch_run(reps=20, timed=False)
dev = hip.batch_run(reps=200, timed=True) / 200 * 1e3
st = hip.persistent_stats() if hasattr(hip.lib, "ltpl_tick_persistent_stats") else {"enabled": 0}
dus, ok = bench.dropin_latency(hip, lat, 1200)
print("[%s] single tick p50 %.1f p99 %.1f mean %.1f | device %.1f | dropin p50 %.1f p99 %.1f keys_ok %s%s" % (
    os.environ["LTPL_AB_TAG"], np.percentile(us, 50), np.percentile(us, 99), us.mean(), dev, np.percentile(dus, 50), np.percentile(dus, 99), ok,
    (" | resident kernel: device %.1f us per tick, %d ticks, %d starts" % (st["device_us_mean"], st["ticks"], st["launches"])) if st["enabled"] else ""))
hip.close()
PY
done
done
