"""
Tick-level recordings of the reference's closed loop (tests/golden/*_ticks.npz, recorded from the unmodified reference by
oracle/gen_golden.py) as INPUT STREAMS and EXPECTED OUTPUTS: conversion of a recorded tick into the arguments of the planner / fleet entry
points, and the comparison of a planner's outputs with what the reference produced on that tick. Shared by the parity tests
(tests/planner_replay.py drives whole closed loops with it) and by bench.py's closed-loop legs (which replay the recorded C2 loop and
spot-check the result) -- the benchmark does not depend on the test tree.

Tolerances (north_star): indices bit-exact, floats 1e-5 relative, every quantity against ITS OWN scale.
"""
import numpy as np

REL_TOL = 1e-5          # north_star: spline coefficients and velocity profiles within 1e-5 relative
ELEM_TOL_VX = 1e-5      # ELEMENT-WISE relative bound on vx where |vx| >= 1 m/s: north_star's "velocity profiles within 1e-5 relative", sample by sample
ELEM_TOL_AX = 1e-5      # the same for ax where |ax| >= 0.5 m/s^2 (ax differentiates v^2: below the floor the difference of two rounded squares decides)
ELEM_FLOOR_VX, ELEM_FLOOR_AX = 1.0, 0.5
KAPPA_FLOOR = 1e-4      # 1/m: curvature magnitudes below 1 / (10 km) are indistinguishable for the planner (the lateral limit
                        # ay / |kappa| is capped by v_max^2 long before); keeps a relative test meaningful on straights: the
                        # absolute tolerance on a path that is straight throughout is 1e-5 * 1e-4 = 1e-9 1/m



def assert_close_rel(actual, desired, rel=REL_TOL, what="", floor=1e-12):
    """max |a - d| <= rel * max(|d|, floor): relative to the magnitude of the reference array (no element-wise blow-up
    at zero crossings)."""
    actual, desired = np.asarray(actual, dtype=float), np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, "%s: shape %s vs %s" % (what, actual.shape, desired.shape)
    if desired.size == 0:
        return
    scale = max(float(np.max(np.abs(desired))), floor)
    err = float(np.max(np.abs(actual - desired)))
    assert err <= rel * scale, "%s: max abs err %.3e > %.1e * %.3e" % (what, err, rel, scale)



def assert_xy_close(actual, desired, rel=REL_TOL, what=""):
    """Coordinates: relative to the EXTENT of the reference path (max - min per column, at least 1 m), not to the magnitude of the
    track coordinates themselves -- a path that is 200 m long at x ~ 1000 m is held to 2 mm, not to 1 cm."""
    actual, desired = np.asarray(actual, dtype=float), np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, "%s: shape %s vs %s" % (what, actual.shape, desired.shape)
    if desired.size == 0:
        return
    for c in range(desired.shape[1]):
        scale = max(float(np.ptp(desired[:, c])), 1.0)
        err = float(np.max(np.abs(actual[:, c] - desired[:, c])))
        assert err <= rel * scale, "%s col %d: max abs err %.3e > %.1e * %.3e" % (what, c, err, rel, scale)



def assert_coeff_close(actual, desired, rel=REL_TOL, what=""):
    """Spline coefficients (rows [a0x a1x a2x a3x a0y a1y a2y a3y], calc_splines.py): every coefficient ORDER against its own scale.
    a0 (knot coordinates) against the extent of the path like ``assert_xy_close``; a1, a2, a3 each against the largest magnitude of
    that order over both axes of the path (an array-wide scale would let a2 / a3 ~ 0.1 .. 1 pass with the absolute error allowed for
    coordinates ~ 10^2 .. 10^3 m). Floors: 1 m for a0, 1e-3 m for the higher orders (a straight segment has a2 = a3 = 0)."""
    actual, desired = np.asarray(actual, dtype=float), np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, "%s: shape %s vs %s" % (what, actual.shape, desired.shape)
    if desired.size == 0:
        return
    assert_xy_close(actual[:, [0, 4]], desired[:, [0, 4]], rel, what + " a0")
    for order in (1, 2, 3):
        cols = [order, 4 + order]
        scale = max(float(np.max(np.abs(desired[:, cols]))), 1e-3)
        err = float(np.max(np.abs(actual[:, cols] - desired[:, cols])))
        assert err <= rel * scale, "%s a%d: max abs err %.3e > %.1e * %.3e" % (what, order, err, rel, scale)



def assert_elementwise(actual, desired, floor, tol, what=""):
    """|a - d| <= tol * |d| for EVERY sample with |d| >= floor. The array-level bounds above are relative to the array's largest value,
    which would let the slow tail of a profile that brakes to standstill drift by far more than 1e-5 of ITS values; this one does not.
    (Round 6: every operand of the velocity stage is fp64 -- measured element-wise error ~1e-13; rounds 3-5 read |kappa| and the element
    lengths as fp32 and needed 1e-4 here.)"""
    actual, desired = np.asarray(actual, dtype=float), np.asarray(desired, dtype=float)
    m = np.abs(desired) >= floor
    if m.any():
        e = float(np.max(np.abs(actual[m] - desired[m]) / np.abs(desired[m])))
        assert e <= tol, "%s element-wise: %.3e > %.0e" % (what, e, tol)



def vehicles_of_tick(t):
    out = []
    for k in range(len(t['obj_radius'])):
        pos = np.asarray(t['obj_pos'][k], dtype=float).reshape(1, 2)
        pred = np.asarray(t['obj_pred'][k], dtype=float).reshape(-1, 2)
        out.append((float(t['obj_radius'][k]), float(t['obj_vel'][k]), np.vstack((pos, pred))))
    return out



def zone_gids_of_tick(lat, t):
    gids = []
    for l, n in zip(t.get('zone_layers', ()), t.get('zone_nodes', ())):
        l, n = int(l), int(n)
        if 0 <= l < lat.num_layers and 0 <= n < lat.nodes_in_layer[l]:
            gids.append(int(lat.layer_off[l]) + n)
    return sorted(set(gids))



def check_traj(got, exp, what):
    assert got.shape == exp.shape, "%s: shape %s vs %s" % (what, got.shape, exp.shape)
    if exp.shape[0] == 0:
        return
    for col, name in enumerate(("s", "x", "y", "psi", "kappa", "vx", "ax")):
        if name == "psi":
            d = np.abs(np.mod(got[:, col] - exp[:, col] + np.pi, 2 * np.pi) - np.pi)
            assert float(d.max()) <= REL_TOL * np.pi, "%s psi" % what
        elif name == "ax":
            scale = max(float(np.max(np.abs(exp[:, 5]))) ** 2 / 2.0, 5.0)
            assert float(np.max(np.abs(got[:, col] - exp[:, col]))) <= 1e-5 * scale, "%s ax: %g" % (
                what, float(np.max(np.abs(got[:, col] - exp[:, col]))))
        elif name == "kappa":
            assert_close_rel(got[:, col], exp[:, col], what="%s kappa" % what, floor=KAPPA_FLOOR)
        elif name in ("x", "y"):
            assert_xy_close(got[:, col:col + 1], exp[:, col:col + 1], what="%s %s" % (what, name))
        else:
            assert_close_rel(got[:, col], exp[:, col], what="%s %s" % (what, name))
            if name == "vx":
                assert_elementwise(got[:, col], exp[:, col], ELEM_FLOOR_VX, ELEM_TOL_VX, "%s vx" % what)
    assert_elementwise(got[:, 6], exp[:, 6], ELEM_FLOOR_AX, ELEM_TOL_AX, "%s ax" % what)



def check_trajectories(traj, ids, ref, t, what):
    """Outputs of get_ref_idx + calc_vel_profile of one tick against the recording (digests every tick, full arrays on selected ticks)."""
    full = t['full']
    er = t['ref_idx']
    assert ref['cut_index_pos'] == er['cut_index_pos'] and ref['cut_layer'] == er['cut_layer'], \
        "%s: cut (%d, %d) vs (%d, %d)" % (what, ref['cut_index_pos'], ref['cut_layer'], er['cut_index_pos'], er['cut_layer'])
    assert abs(ref['vel_plan'] - er['vel_plan']) <= 1e-5 * max(abs(er['vel_plan']), 1.0), "%s: vel_plan" % what
    assert abs(ref['acc_plan'] - er['acc_plan']) <= 1e-5 * max(abs(er['acc_plan']), 5.0), "%s: acc_plan" % what
    assert ref['vel_course'].shape == er['vel_course'].shape, "%s: vel_course length" % what
    if er['vel_course'].size:
        assert np.max(np.abs(ref['vel_course'] - er['vel_course'])) <= 1e-5 * max(float(np.max(np.abs(er['vel_course']))), 1.0)
    ev = t['vel']
    assert list(traj.keys()) == ev['keys'], "%s: trajectory keys %s vs %s" % (what, list(traj.keys()), ev['keys'])
    assert ids == ev['traj_id'], "%s: trajectory ids" % what
    for k in ev['keys']:
        dg = ev['digest'][k]
        tr = traj[k][0]
        assert tr.shape[0] == dg[0], "%s/%s: trajectory rows %d vs %d" % (what, k, tr.shape[0], dg[0])
        vs = max(abs(dg[4]) / max(dg[0], 1), 1.0)
        assert abs(tr[-1, 0] - dg[1]) <= 1e-5 * max(abs(dg[1]), 1.0), "%s/%s: s_end" % (what, k)
        assert abs(tr[0, 5] - dg[2]) <= 1e-5 * max(vs, abs(dg[2])), "%s/%s: vx[0] %g vs %g" % (what, k, tr[0, 5], dg[2])
        assert abs(tr[-1, 5] - dg[3]) <= 1e-5 * max(vs, abs(dg[3])), "%s/%s: vx[-1]" % (what, k)
        assert abs(float(np.sum(tr[:, 5])) - dg[4]) <= 1e-5 * max(abs(dg[4]), 1.0), "%s/%s: sum vx %g vs %g" % (
            what, k, float(np.sum(tr[:, 5])), dg[4])
    if full is not None:
        for k in ev['keys']:
            check_traj(traj[k][0], full['traj'][k], "%s/%s" % (what, k))



def check_digests(dig, t, key_id_of, what=""):
    """Rows of ``Fleet.digest()`` (one per planner, all of which replayed the SAME recording) against tick ``t`` of that recording: what
    ``check_trajectories`` checks on every tick (cut indices, velocity plan, keys / trajectory ids / path ids, rows, s_end, vx[0], vx[-1],
    sum of vx: indices exact, floats 1e-5 relative), for all planners at once. ``key_id_of``: action name -> key id of the ABI
    (planner.KEY_NAMES inverted). Returns the number of planners checked; raises AssertionError naming the first planner that differs."""
    dig = np.asarray(dig, dtype=float)
    if dig.shape[0] == 0:
        return 0
    K = (dig.shape[1] - 8) // 9
    er, ev = t['ref_idx'], t['vel']

    def bad(mask, msg):
        if np.any(mask):
            q = int(np.argmax(mask))
            raise AssertionError("%s planner row %d: %s (digest %s)" % (what, q, msg, dig[q, :8 + 7 * K].tolist()))

    bad(dig[:, 0] != 0, "error word set")
    bad((dig[:, 1] != er['cut_index_pos']) | (dig[:, 2] != er['cut_layer']), "cut (%d, %d)" % (er['cut_index_pos'], er['cut_layer']))
    bad(np.abs(dig[:, 5] - er['vel_plan']) > REL_TOL * max(abs(er['vel_plan']), 1.0), "vel_plan %g" % er['vel_plan'])
    bad(np.abs(dig[:, 7] - er['acc_plan']) > REL_TOL * max(abs(er['acc_plan']), 5.0), "acc_plan %g" % er['acc_plan'])
    bad(dig[:, 6] != er['vel_course'].shape[0], "vel_course length %d" % er['vel_course'].shape[0])
    keys = list(ev['keys'])
    bad(dig[:, 3] != len(keys), "trajectory keys %s" % keys)
    for k, name in enumerate(keys):
        d = dig[:, 8 + 7 * k: 8 + 7 * k + 7]
        dg = ev['digest'][name]
        bad(d[:, 0] != key_id_of[name], "key %d is not %s" % (k, name))
        bad(d[:, 2] != dg[0], "%s: rows %d" % (name, dg[0]))
        vs = max(abs(dg[4]) / max(dg[0], 1), 1.0)
        bad(np.abs(d[:, 3] - dg[1]) > REL_TOL * max(abs(dg[1]), 1.0), "%s: s_end %g" % (name, dg[1]))
        bad(np.abs(d[:, 4] - dg[2]) > REL_TOL * max(vs, abs(dg[2])), "%s: vx[0] %g" % (name, dg[2]))
        bad(np.abs(d[:, 5] - dg[3]) > REL_TOL * max(vs, abs(dg[3])), "%s: vx[-1] %g" % (name, dg[3]))
        bad(np.abs(d[:, 6] - dg[4]) > REL_TOL * max(abs(dg[4]), 1.0), "%s: sum vx %g" % (name, dg[4]))
    # action_set_path_id (OTH.py:696-697,1034): one id per key of the tick INCLUDING dropped keys, in the dict's order
    ids = list(ev['traj_id'].items())
    bad(dig[:, 4] != len(ids), "path ids %s" % ids)
    for k, (name, val) in enumerate(ids[:K]):
        bad((dig[:, 8 + 7 * K + 2 * k] != key_id_of[name]) | (dig[:, 8 + 7 * K + 2 * k + 1] != val), "id %d is not %s: %s" % (k, name, val))
    return int(dig.shape[0])
