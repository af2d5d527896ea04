"""
The FUSED TICK (ltpl_tick_batch / ltpl_batch_run -- the path bench.py times -- and oracle_tick_batch, its checker) against numbers
computed by the UNMODIFIED reference: tests/golden/fresh_ticks.npz (oracle/gen_golden_fresh.py) holds, for 136 scenarios, the
paths of the reference's main_online_path_gen and the trajectories its own OnlineTrajectoryHandler.calc_vel_profile (OTH.py:603-1040)
produces for them with cut_index_pos = 0, cut_layer = 0 and an empty vel_course, i.e. the fused tick's definition
(include/ltpl_hip.h). Covered: all four primitives, three-key templates, reduced-horizon straight AND follow (v_end = 0, zeros on the
last 5 m, row-5 choice OTH.py:923), standstill (ax = -5, OTH.py:938), velocity-bound violations (left / right dropped, OTH.py:943-1015),
four velocity parameter sets (local gg, safety distance, machine limits, v_max).

  CPU   oracle_tick_batch                                                       (pins bench.py's parity checker)
  GPU   ltpl_tick_batch as one batch per parameter set (one-wave batch kernels + lane velocity kernels), one scenario per call
        (fused four-wave k_tick) and the device-resident form ltpl_batch_upload / _run / _download (the timed path of bench.py)
"""
import numpy as np
import pytest

from helpers import load_golden, assert_close_rel, assert_xy_close, assert_vx_elementwise, assert_ax_elementwise, REL_TOL, KAPPA_FLOOR
from graphbasedlocaltrajectoryplanner_amd import _capi

PARAM_KEYS = ("gg", "safety_d", "ax_max_machines", "vel_max")


def records():
    return load_golden("fresh_ticks.npz")


def scenario_of(rec):
    sc = rec["scen"]
    return {"start_node": tuple(sc["start_node"]), "action_sets": True,
            "vehicles": [(float(r), np.asarray(p, dtype=float)) for r, p in zip(sc["veh_radius"], sc["veh_pos"])],
            "zone_gids": list(sc["zone_gids"]), "last_nodes": sc["last_nodes"], "obj_in_const": False, "obj_besides": False,
            "last_action": None, "const_closest": None, "psi_s": None}


def pack(lat, recs):
    """(PathsBatch, TickVelBatch) of records that share one velocity parameter set."""
    vi = recs[0]["vel_in"]
    params = _capi.VelParamSet(len_veh=lat.veh_length, v_max=vi["vel_max"], ax_max_machines=vi["ax_max_machines"])
    batch = _capi.PathsBatch([scenario_of(r) for r in recs], w_last_edges=recs[0]["scen"]["w_last_edges"])
    veh_vel = np.concatenate([np.asarray(r["scen"]["veh_vel"], dtype=float).reshape(-1) for r in recs] + [np.zeros(0)])
    vel = _capi.TickVelBatch(params, len(recs), [r["vel_in"]["vel_plan"] for r in recs], [r["vel_in"]["vel_est"] for r in recs],
                             np.array([r["vel_in"]["pos_est"] for r in recs]), veh_vel, gg=tuple(vi["gg"]),
                             safety_d=vi["safety_d"], v_max_offset=0.1)
    return batch, vel


def check_record(lat, res, vres, i, rec, what):
    """Scenario i of a fused-tick result against the reference's record."""
    exp_p, exp_v = rec["paths"], rec["vel"]
    nodes, node_idx, coeff, path_param, red_len, closest = res.action_sets(i, rec["scen"]["start_node"][0], lat.num_layers)
    assert list(nodes.keys()) == exp_p["keys"], "%s: keys %s vs %s" % (what, list(nodes.keys()), exp_p["keys"])
    assert closest == exp_p["closest_obj_index"], "%s: closest object" % what
    slot_of = {}
    for a in range(int(res.n_actions[i])):
        if res.valid[i, a]:
            slot_of[_capi.ACTION_NAMES[int(res.action_id[i, a])]] = a
    for k in exp_p["keys"]:
        a = slot_of[k]
        assert nodes[k][0] == exp_p["nodes"][k], "%s/%s: node list" % (what, k)                      # bit-exact
        assert red_len[k][0] == exp_p["red_len"][k], "%s/%s: reduced flag" % (what, k)
        n = exp_p["n_rows"][k]
        assert path_param[k][0].shape[0] == n
        if k in exp_v["dropped"]:
            assert int(vres.vel_bound[i, a]) == 0, "%s/%s: the reference dropped this primitive (velocity bound)" % (what, k)
            continue
        tr = exp_v["traj"][k]                                    # [s, x, y, psi, kappa, vx, ax]
        assert tr.shape == (n, 7)
        pp = path_param[k][0]
        assert_xy_close(pp[:, 0:2], tr[:, 1:3], what="%s/%s xy" % (what, k))
        dpsi = np.abs(np.mod(pp[:, 2] - tr[:, 3] + np.pi, 2 * np.pi) - np.pi)
        assert float(dpsi.max()) <= REL_TOL * np.pi, "%s/%s psi" % (what, k)
        assert_close_rel(pp[:, 3], tr[:, 4], what="%s/%s kappa" % (what, k), floor=KAPPA_FLOOR)
        s = np.concatenate(([0.0], np.cumsum(pp[:-1, 4])))
        assert_close_rel(s, tr[:, 0], rel=1e-12, what="%s/%s s" % (what, k))
        vx, ax = vres.vx[i, a, :n], vres.ax[i, a, :n]
        assert_close_rel(vx, tr[:, 5], what="%s/%s vx" % (what, k), floor=1.0)
        scale = max(float(np.max(np.abs(tr[:, 5]))) ** 2 / 2.0, 5.0)   # ax = d(v^2) / (2 ds): against the scale of v^2 / ds
        err = float(np.max(np.abs(ax - tr[:, 6])))
        assert err <= 1e-5 * scale, "%s/%s ax: %.3e" % (what, k, err)
        assert_vx_elementwise(vx, tr[:, 5], "%s/%s" % (what, k))
        assert_ax_elementwise(ax, tr[:, 6], "%s/%s" % (what, k))
        # standstill rule (OTH.py:938): exactly -5 where the reference has it
        assert np.array_equal(ax == -5.0, tr[:, 6] == -5.0), "%s/%s: standstill rows" % (what, k)
        if k in ("left", "right"):
            assert int(vres.vel_bound[i, a]) == 1, "%s/%s: kept by the reference, so the velocity bound holds" % (what, k)
        else:
            vb = abs(tr[0, 5] - rec["vel_in"]["vel_plan"]) < 0.1 if (k != "follow" or exp_p["red_len"][k]) else exp_v["follow"][1]
            assert bool(vres.vel_bound[i, a]) == bool(vb), "%s/%s: vel_bound" % (what, k)
        if k == "follow" and exp_v["follow"] is not None:
            assert bool(vres.too_close[i, a]) == bool(exp_v["follow"][0]), "%s/follow: too_close" % what


def groups(recs):
    by = {}
    for j, r in enumerate(recs):
        by.setdefault(r["param_set"], []).append(j)
    return by


def run_batched(lat, backend, recs, resident=False):
    seen = {"reduced": 0, "dropped": 0, "three": 0, "standstill": 0, "keys": set()}
    for ps, idx in sorted(groups(recs).items()):
        sub = [recs[j] for j in idx]
        batch, vel = pack(lat, sub)
        if resident:
            backend.batch_upload(batch, vel)
            backend.batch_run(reps=2, timed=False)
            res, vres = backend.batch_download()
        else:
            res, vres = backend.tick_batch(batch, vel)
        for i, (j, rec) in enumerate(zip(idx, sub)):
            check_record(lat, res, vres, i, rec, "record %d (set %d)" % (j, ps))
            seen["reduced"] += int(any(rec["paths"]["red_len"].values()))
            seen["dropped"] += len(rec["vel"]["dropped"])
            seen["three"] += int(len(rec["paths"]["keys"]) == 3)
            seen["standstill"] += int(rec["vel_in"]["vel_plan"] == 0.0)
            seen["keys"].update(rec["paths"]["keys"])
    return seen


def test_fixture_covers_the_branches():
    recs = records()
    assert len(recs) >= 120 and len(groups(recs)) == 4
    red = [r for r in recs if any(r["paths"]["red_len"].values())]
    assert any("follow" in r["paths"]["keys"] for r in red) and any("straight" in r["paths"]["keys"] for r in red)
    assert sum(len(r["vel"]["dropped"]) for r in recs) >= 10
    assert sum(len(r["paths"]["keys"]) == 3 for r in recs) >= 5
    assert any((r["vel"]["traj"][k][:, 6] == -5.0).any() for r in recs for k in r["vel"]["keys"])
    assert any(r["vel"]["follow"] is not None and r["vel"]["follow"][0] for r in recs)            # too_close occurs


def test_oracle_tick_batch_matches_the_reference(monteblanco, oracle_backend):
    seen = run_batched(monteblanco, oracle_backend, records())
    assert seen["keys"] == {"straight", "follow", "left", "right"} and seen["reduced"] >= 8 and seen["dropped"] >= 10


@pytest.mark.gpu
def test_hip_tick_batch_matches_the_reference(monteblanco, hip_backend):
    seen = run_batched(monteblanco, hip_backend, records())
    assert seen["keys"] == {"straight", "follow", "left", "right"} and seen["reduced"] >= 8 and seen["dropped"] >= 10


@pytest.mark.gpu
def test_hip_resident_batch_matches_the_reference(monteblanco, hip_backend):
    """ltpl_batch_upload / _run / _download: the device-resident pipeline inside bench.py's timed region."""
    run_batched(monteblanco, hip_backend, records(), resident=True)


@pytest.mark.gpu
def test_hip_single_tick_matches_the_reference(monteblanco, hip_backend):
    """One scenario per call: the fused four-wave kernel (k_tick) of the latency path."""
    recs = records()
    for j, rec in enumerate(recs):
        batch, vel = pack(monteblanco, [rec])
        res, vres = hip_backend.tick_batch(batch, vel)
        check_record(monteblanco, res, vres, 0, rec, "record %d (single)" % j)
