"""Spill study: run ONE library variant (LTPL_HIP_LIB) with the runtime LDS plan forced, so that the
one-wave batch kernel k_paths<1, PlanRt> -- built with a 128-VGPR budget in the lib_spill* variants and therefore spilling -- plans
a batch of C3 scenarios (symmetric oval: exact cost ties on most layers); compare with the oracle and with the four-wave kernel.
  usage: LTPL_HIP_LIB=... python tools/ubench/spill_study/run.py [n_scen]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ["LTPL_NO_FIXED_PLAN"] = "1"
from graphbasedlocaltrajectoryplanner_amd import _capi                                    # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice, scattered_obstacle_scenarios   # noqa: E402
from oracle.oracle_lib import OracleBackend                                               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lat = c3_lattice()
scen, _ = scattered_obstacle_scenarios(lat, n, n_obj=32, seed=3)
batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
ref = OracleBackend(lat).plan_paths(batch)
out = {"lib": os.path.basename(os.environ.get("LTPL_HIP_LIB", "default"))}
for selftest in (True, False):
    os.environ.pop("LTPL_NO_SELFTEST", None)
    if not selftest:
        os.environ["LTPL_NO_SELFTEST"] = "1"
    try:
        hip = _capi.HipBackend(lat)
    except _capi.BackendError as e:
        out["create_with_selftest"] = "REFUSED: " + str(e)[-90:]
        continue
    if selftest:
        out["create_with_selftest"] = "accepted"
    res = hip.plan_paths(batch)                                  # >= 64 scenarios: one-wave batch kernel
    bad = 0
    for s in range(n):
        same = np.array_equal(res.valid[s], ref.valid[s]) and np.array_equal(res.n_nodes[s] * res.valid[s], ref.n_nodes[s] * ref.valid[s])
        if same:
            for a in range(3):
                if ref.valid[s, a] and not np.array_equal(res.nodes[s, a, :ref.n_nodes[s, a]], ref.nodes[s, a, :ref.n_nodes[s, a]]):
                    same = False
        bad += 0 if same else 1
    out["batch_kernel_vs_oracle_mismatching_scenarios"] = "%d / %d" % (bad, n)
    small = _capi.PathsBatch(scen[:48], w_last_edges=[0.0, 0.5, 0.8])
    r4, o4 = hip.plan_paths(small), OracleBackend(lat).plan_paths(small)       # < 64 scenarios: four-wave kernel
    ok4 = np.array_equal(r4.valid, o4.valid)
    for s in range(48):
        for a in range(3):
            if o4.valid[s, a] and r4.valid[s, a]:
                ok4 = ok4 and np.array_equal(r4.nodes[s, a, :o4.n_nodes[s, a]], o4.nodes[s, a, :o4.n_nodes[s, a]])
    out["four_wave_kernel_vs_oracle"] = "ok" if ok4 else "MISMATCH"
    hip.close()
    break
print(out)
