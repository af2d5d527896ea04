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


def test_previous_solution_discount_with_any_alignment(monteblanco, hip_backend, oracle_backend):
    """last_solution_nodes need not start on the start layer (GraphBase.factor_edge_cost discounts whatever edges the node pairs
    span, GraphBase.py:478-512): shifted lists must give the oracle's paths, in the batch kernel and in the four-wave kernel."""
    from scenarios import random_scenarios
    lat = monteblanco
    L = lat.num_layers
    scen, _ = random_scenarios(lat, 96, seed=4242, last_prob=1.0)
    rng = np.random.default_rng(7)
    for i, sc in enumerate(scen):
        sl, sn = sc["start_node"]
        d = int(rng.integers(0, 4)) if i % 3 else 0           # two thirds start 1..3 layers behind the start node
        nodes, cur = [], sn
        for j in range(0, d + 5):
            l2 = (sl + j) % L
            if j > 0:
                cands = [nn for nn in range(lat.nodes_in_layer[l2]) if lat.find_edge((l2 - 1) % L, cur, l2, nn) >= 0]
                if not cands:
                    break
                cur = int(cands[int(rng.integers(0, len(cands)))])
            nodes.append([l2, cur])
        sc["last_nodes"] = nodes[d:]
    for chunk in (scen, scen[:7]):                             # >= 64 scenarios: batch kernel; fewer: four-wave kernel
        batch = _capi.PathsBatch(chunk, w_last_edges=[0.0, 0.5, 0.8])
        res, ref = hip_backend.plan_paths(batch), oracle_backend.plan_paths(batch)
        for name in ("n_actions", "action_id", "valid", "reduced", "n_nodes", "n_pts"):
            assert np.array_equal(getattr(res, name), getattr(ref, name)), name
        for s in range(len(chunk)):
            for a in range(int(ref.n_actions[s])):
                if ref.valid[s, a]:
                    nn = int(ref.n_nodes[s, a])
                    assert np.array_equal(res.nodes[s, a, :nn], ref.nodes[s, a, :nn]), (s, a)


def test_resident_batch_is_dropped_by_other_entry_points(monteblanco, hip_backend):
    """ltpl_batch_run after another entry point reused the staging buffers must fail loudly, not read stale memory."""
    from scenarios import random_scenarios
    scen, vels = random_scenarios(monteblanco, 64, seed=5)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    params = _capi.VelParamSet(len_veh=monteblanco.veh_length)
    pos = np.array([monteblanco.node_pos[monteblanco.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    vel = _capi.TickVelBatch(params, 64, np.full(64, 20.0), np.full(64, 20.0), pos, np.concatenate(vels))
    hip_backend.batch_upload(batch, vel)
    hip_backend.batch_run(reps=1, timed=False)
    hip_backend.plan_paths(_capi.PathsBatch(scen[:3], w_last_edges=[0.0, 0.5, 0.8]))
    with pytest.raises(_capi.BackendError):
        hip_backend.batch_run(reps=1, timed=False)


def test_obstacles_off_the_closest_layer_grid(monteblanco, hip_backend, oracle_backend, monkeypatch):
    """Phase 1 of the path kernel looks the closest reference-line layer of an obstacle position up in a create-time grid (round 5). Positions
    OUTSIDE the grid (far off the track), in its far corners (cells marked "full scan": the middle of the circuit) and NaN-free extremes make
    the scenario take the scan over all layers -- results must equal the oracle's full argmin either way, in the one-wave and the four-wave
    kernel, also when only ONE of many positions is off the grid; and a handle created without the grid (LTPL_NO_LAYER_GRID=1) must agree
    bit for bit with the default one on ordinary scenarios."""
    lat = monteblanco
    rng = np.random.default_rng(5)
    cx, cy = float(np.mean(lat.refline[:, 0])), float(np.mean(lat.refline[:, 1]))
    far = [np.array([[cx, cy], [cx, cy]]),                                                  # middle of the circuit: many equally distant layers
           np.array([[float(lat.refline[:, 0].max()) + 500.0, cy], [float(lat.refline[:, 0].max()) + 500.0, cy]]),      # outside the grid
           np.array([[-1.0e7, 3.0e6], [-1.0e7, 3.0e6]])]                                    # very far outside
    scen = crowded(lat, 72, n_veh=6, n_pred=1, seed=31)
    for i, sc in enumerate(scen):
        if i % 3 == 0:
            sc["vehicles"] = sc["vehicles"] + [(2.0, far[(i // 3) % 3])]                     # one off-grid vehicle among ordinary ones
        elif i % 3 == 1:
            sc["vehicles"] = [(2.0, far[j]) for j in range(3)]                              # only off-grid vehicles
    for group in (scen, scen[:6]):                                                          # one-wave teams / four-wave teams
        batch = _capi.PathsBatch(group, w_last_edges=W_LAST)
        compare_results(hip_backend.plan_paths(batch), oracle_backend.plan_paths(batch), lat)
    monkeypatch.setenv("LTPL_NO_LAYER_GRID", "1")
    plain = _capi.HipBackend(lat)
    ordinary = crowded(lat, 80, n_veh=8, n_pred=1, seed=32)
    batch = _capi.PathsBatch(ordinary, w_last_edges=W_LAST)
    a, b = hip_backend.plan_paths(batch), plain.plan_paths(batch)
    compare_results(a, b, lat)
    for i in range(len(ordinary)):                            # (entries behind n_actions / n_nodes / n_pts are unspecified padding)
        na = int(a.n_actions[i])
        assert na == int(b.n_actions[i]) and int(a.closest_obj_index[i]) == int(b.closest_obj_index[i])
        for k in range(na):
            assert int(a.valid[i, k]) == int(b.valid[i, k]) and int(a.action_id[i, k]) == int(b.action_id[i, k])
            if a.valid[i, k]:
                nn, npts = int(a.n_nodes[i, k]), int(a.n_pts[i, k])
                assert nn == int(b.n_nodes[i, k]) and npts == int(b.n_pts[i, k])
                assert np.array_equal(a.nodes[i, k, :nn], b.nodes[i, k, :nn]) and np.array_equal(a.path_param[i, k, :npts], b.path_param[i, k, :npts])
    plain.close()


@pytest.mark.parametrize("env", [{"LTPL_POLL": "1"}, {"LTPL_ZC_OUT": "0"}, {"LTPL_ZC_IN": "1", "LTPL_POLL": "1"}, {"LTPL_TICK_GRAPH": "1"},
                                 {"LTPL_TICK_GRAPH": "1", "LTPL_ZC_OUT": "0"},
                                 {"LTPL_PERSISTENT_TICK": "1", "LTPL_PERSIST_IDLE_MS": "50"}])     # (round 6: the resident tick kernel, by environment)
def test_latency_transport_variants_give_identical_results(monteblanco, hip_backend, monkeypatch, env):
    """The transport of the single-tick path (zero-copy outputs in page-locked memory, optional polled completion word, optional
    zero-copy inputs) must not change a bit of the results: a second handle created under the switches against the default one,
    on seam (1), seam (2) jobs through the fused tick, one scenario per call."""
    from scenarios import random_scenarios
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    other = _capi.HipBackend(monteblanco)
    scen, vels = random_scenarios(monteblanco, 12, seed=77)
    params = _capi.VelParamSet(len_veh=monteblanco.veh_length)
    for i in range(12):
        batch = _capi.PathsBatch(scen[i:i + 1], w_last_edges=W_LAST)
        a, b = hip_backend.plan_paths(batch), other.plan_paths(batch)
        assert int(a.n_actions[0]) == int(b.n_actions[0])
        na = int(a.n_actions[0])
        for name in ("action_id", "valid", "n_nodes", "n_pts"):
            assert np.array_equal(getattr(a, name)[0, :na], getattr(b, name)[0, :na]), (i, name)
        for k in range(na):
            if a.valid[0, k]:
                n, nn = int(a.n_pts[0, k]), int(a.n_nodes[0, k])
                assert np.array_equal(a.nodes[0, k, :nn], b.nodes[0, k, :nn]) and np.array_equal(a.node_idx[0, k, :nn], b.node_idx[0, k, :nn])
                assert np.array_equal(a.path_param[0, k, :n], b.path_param[0, k, :n]), (i, k)
        sl, sn = scen[i]['start_node']
        pos = monteblanco.node_pos[monteblanco.layer_off[sl] + sn][None, :]
        vel = _capi.TickVelBatch(params, 1, np.full(1, 20.0), np.full(1, 20.0), pos, vels[i])
        ra, va = hip_backend.tick_batch(batch, vel)
        rb, vb = other.tick_batch(batch, vel)
        for k in range(int(ra.n_actions[0])):
            if ra.valid[0, k]:
                n = int(ra.n_pts[0, k])
                assert np.array_equal(va.vx[0, k, :n], vb.vx[0, k, :n]) and np.array_equal(va.ax[0, k, :n], vb.ax[0, k, :n]), (i, k)
        assert np.array_equal(va.vel_bound, vb.vel_bound) and np.array_equal(va.too_close, vb.too_close)
    other.close()
