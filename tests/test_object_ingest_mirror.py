"""
Container-only (needs /root/reference): the object-ingestion mirror against the REAL
``ObjectListInterface.process_object_list`` in the closed loop.

The C2 opponents are joined, every tick, by objects that exercise each branch of ObjectListInterface.py:75-153:
  * a large static object just OUTSIDE the right bound a little ahead of the ego (dropped by the on-track test; if it were
    kept its 10 m radius would block lattice edges and change the plan),
  * seeded random objects straddling both bounds on a far part of the track (kept / dropped per check_inside_bounds),
  * an object that brings its own 'prediction',
  * an object of an unsupported type (warning, ignored).
Run A = unmodified reference, run B = after install(); the vehicle lists handed to the path seam must be IDENTICAL
(ids, order, position, heading, radius, velocity, prediction -- bit for bit) and the exported trajectories must agree.
No GPU here, so run B binds the mirror to the oracle's C backend (host logic under test; the device arithmetic is checked
against the same oracle and the same reference-generated golden vectors by the ``-m gpu`` tests).
"""
import os
import warnings
import numpy as np
import pytest

from helpers import assert_close_rel, ROOT
from oracle import ref_env

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_env.reference_available(), reason="reference tree not present")]

CACHE = os.path.join(ROOT, "oracle", "_cache")
N_TICKS = 60


def run(patched):
    from oracle import ref_scenarios as rs
    from oracle.oracle_lib import OracleBackend
    from graphbasedlocaltrajectoryplanner_amd.install import install, uninstall
    warnings.simplefilter("ignore")
    session = None
    if patched:
        graph_ltpl, _ = ref_env.load_reference()
        session = install(graph_ltpl, backend_factory=OracleBackend)
    try:
        gl, clock, ltpl_obj, gb, path_dict = rs.make_planner(CACHE)
        refline, normvec = np.asarray(gb.refline), np.asarray(gb.normvec_normalized)
        w_r, w_l = np.asarray(gb.track_width_right), np.asarray(gb.track_width_left)
        n = refline.shape[0]
        rng = np.random.default_rng(11)

        def extras(tick):
            objs = []
            i = 6                                                   # ~ a few layers ahead of the start position
            p = refline[i] + normvec[i] * (w_r[i] + 2.0)
            objs.append({'id': 900, 'type': 'physical', 'X': float(p[0]), 'Y': float(p[1]), 'theta': 0.3, 'v': 0.0,
                         'length': 20.0, 'width': 2.0})
            for k in range(6):                                      # straddling the bounds, far from the ego
                i = int(rng.integers(n // 2, n // 2 + 30))
                side = 1.0 if k % 2 else -1.0
                w = w_r[i] if side > 0 else w_l[i]
                p = refline[i] + normvec[i] * side * (w + rng.uniform(-0.6, 0.6))
                objs.append({'id': 910 + k, 'type': 'physical', 'X': float(p[0]), 'Y': float(p[1]),
                             'theta': float(rng.uniform(-3, 3)), 'v': float(rng.uniform(0, 40)), 'length': 4.0,
                             'width': 2.0})
            i = n // 2 + 40
            objs.append({'id': 950, 'type': 'physical', 'X': float(refline[i, 0]), 'Y': float(refline[i, 1]),
                         'theta': 0.0, 'v': 5.0, 'length': 5.0, 'width': 2.0,
                         'prediction': np.array([[refline[i, 0] + 1.0, refline[i, 1]], [refline[i, 0] + 2.0, refline[i, 1]]])})
            objs.append({'id': 960, 'type': 'lidar_blob', 'X': 0.0, 'Y': 0.0})
            return objs

        seen = []

        def on_tick(tick, exported):
            vehs = ltpl_obj._Graph_LTPL__obj_veh
            seen.append([(v.id, np.array(v.get_pos(), float), float(v.get_psi()) if hasattr(v, "get_psi") else None,
                          float(v.get_radius()), float(v.get_vel()), np.array(v.get_prediction(), float))
                         for v in vehs])

        exported = rs.run_loop(gl, clock, ltpl_obj, path_dict, n_ticks=N_TICKS, dt=0.05, dummies=rs.opponents_c2(gl, 8),
                               zones=rs.ZONE_EXAMPLE, on_tick=on_tick, extra_objects=extras)
        return exported, seen
    finally:
        if session is not None:
            uninstall(session)


def test_object_ingestion_mirror_closed_loop():
    ref, ref_seen = run(False)
    got, got_seen = run(True)
    assert len(ref_seen) == len(got_seen) == N_TICKS
    n_dropped = 0
    for t, (a, b) in enumerate(zip(ref_seen, got_seen)):
        assert [v[0] for v in a] == [v[0] for v in b], "tick %d: kept object ids differ" % t
        ids = [v[0] for v in a]
        assert 900 not in ids and 960 not in ids and 950 in ids
        n_dropped += sum(1 for k in range(6) if 910 + k not in ids)
        for va, vb in zip(a, b):
            assert np.array_equal(va[1], vb[1]) and va[2] == vb[2] and va[3] == vb[3] and va[4] == vb[4], "tick %d" % t
            assert va[5].shape == vb[5].shape and np.array_equal(va[5], vb[5]), "tick %d id %d prediction" % (t, va[0])
    assert 0 < n_dropped < 6 * N_TICKS          # the straddling objects exercised both outcomes
    for t, (a, b) in enumerate(zip(ref, got)):
        assert list(a['traj'].keys()) == list(b['traj'].keys()), "tick %d" % t
        for k in a['traj']:
            assert a['traj'][k].shape == b['traj'][k].shape
            assert_close_rel(b['traj'][k][:, 1:3], a['traj'][k][:, 1:3], what="tick %d %s xy" % (t, k))
            assert_close_rel(b['traj'][k][:, 5], a['traj'][k][:, 5], what="tick %d %s vx" % (t, k))
