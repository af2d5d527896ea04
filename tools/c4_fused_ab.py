#!/usr/bin/env python
"""Where is the crossover between the fused tick kernel (one four-wave workgroup per scenario, ONE launch) and the batch pipeline (one wave per
scenario + three velocity kernels) for SMALL batches -- BASELINE config C4's shard is 128 scenarios per GPU. Resident inputs, us per step.
    python tools/c4_fused_ab.py [n ...]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256, 512, 1024, 2048]
lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
scen, batch, vel = bench.make_batch(lat, max(sizes), seed=1)
for rep in range(2):
    for forced in (0, 1):
        if forced:
            os.environ["LTPL_FORCE_FUSED"] = "1"
        else:
            os.environ.pop("LTPL_FORCE_FUSED", None)
        hip = _capi.HipBackend(lat)
        row = []
        for n in sizes:
            b, v = bench.sub_batch(scen, batch, vel, 0, n)
            hip.batch_upload(b, v)
            hip.batch_run(reps=20, timed=False)
            t0 = time.perf_counter(); hip.batch_run(reps=200, timed=True); el = time.perf_counter() - t0
            row.append("%5d: %7.1f us" % (n, el / 200 * 1e6))
        print("%-8s %s" % ("fused" if forced else "default", "  ".join(row)))
        hip.close()
