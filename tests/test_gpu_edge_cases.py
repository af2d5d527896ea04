"""GPU: edge cases of seam (1) / the tick through the C ABI against the oracle -- capacity maxima, degenerate and blocked
scenarios, batch sizes around the one-wave / four-wave team switch ("empty and ragged inputs, maximum
sizes")."""
import numpy as np
import pytest

from test_gpu_paths import compare_results
from scenarios import random_scenarios, raceline_state
from graphbasedlocaltrajectoryplanner_amd import _capi

pytestmark = pytest.mark.gpu
W_LAST = [0.0, 0.5, 0.8]


def crowded(lat, n, n_veh, n_pred, seed):
    """``n_veh`` vehicles with ``n_pred`` prediction points each, all inside the planning horizon of the ego vehicle."""
    rng = np.random.default_rng(seed)
    scen = []
    for _ in range(n):
        sl = int(rng.integers(0, lat.num_layers))
        sn = int(lat.raceline_index[sl])
        vehicles = []
        for _k in range(n_veh):
            x, y, psi, v = raceline_state(lat, float(lat.s_raceline[sl]) + rng.uniform(10.0, 200.0))
            i = int(np.argmin((lat.refline[:, 0] - x) ** 2 + (lat.refline[:, 1] - y) ** 2))
            off = rng.uniform(-5.0, 5.0)
            x, y = x + lat.normvec[i, 0] * off, y + lat.normvec[i, 1] * off
            pts = [[x, y]] + [[x - np.sin(psi) * 3.0 * (m + 1), y + np.cos(psi) * 3.0 * (m + 1)] for m in range(n_pred)]
            vehicles.append((float(rng.uniform(0.5, 3.0)), np.array(pts)))
        scen.append({"start_node": (sl, sn), "action_sets": True, "vehicles": vehicles, "zone_gids": [], "last_nodes": None,
                     "obj_in_const": False, "obj_besides": False, "last_action": None, "const_closest": None, "psi_s": None})
    return scen


def test_capacity_maxima_96_vehicles_192_positions(monteblanco, hip_backend, oracle_backend):
    scen = crowded(monteblanco, 70, n_veh=96, n_pred=1, seed=21)            # 96 vehicles x 2 positions = 192 positions
    batch = _capi.PathsBatch(scen, w_last_edges=W_LAST)
    assert int(batch.pos_off[-1]) == 70 * 192
    compare_results(hip_backend.plan_paths(batch), oracle_backend.plan_paths(batch), monteblanco)       # one-wave teams
    b4 = _capi.PathsBatch(scen[:5], w_last_edges=W_LAST)
    compare_results(hip_backend.plan_paths(b4), oracle_backend.plan_paths(b4), monteblanco)              # four-wave teams
    many_pred = crowded(monteblanco, 66, n_veh=12, n_pred=15, seed=22)      # ragged: 16 positions per vehicle
    bm = _capi.PathsBatch(many_pred, w_last_edges=W_LAST)
    compare_results(hip_backend.plan_paths(bm), oracle_backend.plan_paths(bm), monteblanco)


def test_everything_blocked_and_nothing_blocked(monteblanco, hip_backend, oracle_backend):
    lat = monteblanco
    scen = []
    for sl in (0, 17, 64, 127):
        sn = int(lat.raceline_index[sl])
        l1 = (sl + 1) % lat.num_layers
        wall = [int(lat.layer_off[l1]) + n for n in range(int(lat.nodes_in_layer[l1]))]     # the whole next layer is a zone
        base = {"start_node": (sl, sn), "action_sets": True, "vehicles": [], "last_nodes": None, "obj_in_const": False,
                "obj_besides": False, "last_action": None, "const_closest": None, "psi_s": None}
        scen.append(dict(base, zone_gids=wall))                                             # no primitive can leave the start layer
        scen.append(dict(base, zone_gids=[]))                                               # free track: straight only
        scen.append(dict(base, zone_gids=[int(lat.layer_off[sl]) + sn]))                    # the start node itself is in a zone
        scen.append(dict(base, zone_gids=[], action_sets=False))
    for group in (scen, scen * 5):                                                          # 16 (four-wave) and 80 (one-wave) scenarios
        batch = _capi.PathsBatch(group, w_last_edges=W_LAST)
        res, ref = hip_backend.plan_paths(batch), oracle_backend.plan_paths(batch)
        compare_results(res, ref, lat)
        assert int(res.valid[0].sum()) == 0 and int(res.valid[1].sum()) == 1


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129])
def test_batch_sizes_around_the_team_switch(monteblanco, hip_backend, oracle_backend, n):
    scen, vels = random_scenarios(monteblanco, 129, seed=31, n_veh=6)
    full = _capi.PathsBatch(scen, w_last_edges=W_LAST)
    ref = oracle_backend.plan_paths(full)
    sub = _capi.PathsBatch(scen[:n], w_last_edges=W_LAST)
    res = hip_backend.plan_paths(sub)
    for name in ("valid", "action_id", "n_nodes", "n_pts", "n_ties", "reduced", "goal_layer"):
        assert np.array_equal(getattr(res, name), getattr(ref, name)[:n]), name
    for s in range(n):
        for a in range(3):
            if ref.valid[s, a]:
                nn = int(ref.n_nodes[s, a])
                assert np.array_equal(res.nodes[s, a, :nn], ref.nodes[s, a, :nn])
