"""GPU: seam (1) through the C ABI (libltpl_hip.so) against the reference recordings and against the oracle."""
import numpy as np
import pytest

from helpers import (load_golden, replay_path_call, check_path_output, assert_close_rel, assert_xy_close, assert_coeff_close, REL_TOL,
                     KAPPA_FLOOR)
from scenarios import random_scenarios
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.path_gen import OnlinePathGenerator

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture", ["c2_path_calls.npz", "c1_path_calls.npz", "zonewall_path_calls.npz"])
def test_hip_matches_reference_recordings(monteblanco, hip_backend, fixture):
    recs = load_golden(fixture)
    gen = OnlinePathGenerator(monteblanco, hip_backend)
    scen = [replay_path_call(gen, r) for r in recs]
    # one launch for the whole fixture (one workgroup per recorded call) ...
    batch = _capi.PathsBatch(scen, w_last_edges=recs[0]['w_last_edges'])
    res = hip_backend.plan_paths(batch)
    for i, rec in enumerate(recs):
        check_path_output(res.action_sets(i, rec['start_node'][0], monteblanco.num_layers), rec,
                          what="%s tick %d" % (fixture, rec['tick']))
    # ... and a few single-scenario launches through the same call the drop-in mirror makes
    for rec in recs[::15]:
        b1 = _capi.PathsBatch([replay_path_call(gen, rec)], w_last_edges=rec['w_last_edges'])
        r1 = hip_backend.plan_paths(b1)
        check_path_output(r1.action_sets(0, rec['start_node'][0], monteblanco.num_layers), rec,
                          what="%s tick %d (single)" % (fixture, rec['tick']))


def compare_results(res, ref, lat):
    """HIP vs oracle on identical packed inputs: indices bit-exact, floats within 1e-5 relative (coordinates: of the path's extent;
    spline coefficients: per coefficient order; helpers.py)."""
    for name in ("end_layer", "closest_obj_index", "closest_obj_node", "n_actions", "action_id", "valid", "reduced",
                 "goal_layer", "n_nodes", "n_pts", "n_ties"):
        assert np.array_equal(getattr(res, name), getattr(ref, name)), name
    n_scen = res.n_scen
    for s in range(n_scen):
        for a in range(int(res.n_actions[s])):
            if not res.valid[s, a]:
                continue
            nn, npts = int(res.n_nodes[s, a]), int(res.n_pts[s, a])
            assert np.array_equal(res.nodes[s, a, :nn], ref.nodes[s, a, :nn])
            assert np.array_equal(res.node_idx[s, a, :nn], ref.node_idx[s, a, :nn])
            assert_coeff_close(res.coeff[s, a, :nn - 1], ref.coeff[s, a, :nn - 1], what="coeff s%d a%d" % (s, a))
            pp, rp = res.path_param[s, a, :npts], ref.path_param[s, a, :npts]
            assert_xy_close(pp[:, 0:2], rp[:, 0:2], what="xy s%d a%d" % (s, a))
            dpsi = np.abs(np.mod(pp[:, 2] - rp[:, 2] + np.pi, 2 * np.pi) - np.pi)
            assert float(dpsi.max()) <= REL_TOL * np.pi
            assert_close_rel(pp[:, 3], rp[:, 3], what="kappa s%d a%d" % (s, a), floor=KAPPA_FLOOR)
            assert np.array_equal(pp[:, 4], rp[:, 4])


@pytest.mark.parametrize("seed,n_veh", [(0, 8), (1, 8), (2, 3), (3, 0), (4, 16)])
def test_hip_matches_oracle_on_random_scenarios(monteblanco, hip_backend, oracle_backend, seed, n_veh):
    scen, _ = random_scenarios(monteblanco, 256, seed=seed, n_veh=n_veh)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    res = hip_backend.plan_paths(batch)
    ref = oracle_backend.plan_paths(batch)
    compare_results(res, ref, monteblanco)
    # the scenario mix must actually exercise the branches
    names = set(int(x) for x in res.action_id[res.valid == 1])
    if n_veh >= 3:
        assert _capi.ACT_FOLLOW in names and names & {_capi.ACT_LEFT, _capi.ACT_RIGHT}


def test_scenario_independence_and_determinism(monteblanco, hip_backend):
    """Size-independent property: a scenario's result does not depend on its batch neighbours or on repetition."""
    scen, _ = random_scenarios(monteblanco, 1024, seed=11, n_veh=8)
    w = [0.0, 0.5, 0.8]
    full = hip_backend.plan_paths(_capi.PathsBatch(scen, w_last_edges=w))
    again = hip_backend.plan_paths(_capi.PathsBatch(scen, w_last_edges=w))
    for name in ("nodes", "node_idx", "coeff", "path_param", "valid", "action_id", "n_pts"):
        assert np.array_equal(getattr(full, name), getattr(again, name)), name
    perm = np.random.default_rng(5).permutation(len(scen))[:128]
    sub = hip_backend.plan_paths(_capi.PathsBatch([scen[i] for i in perm], w_last_edges=w))
    for k, i in enumerate(perm):
        assert np.array_equal(sub.valid[k], full.valid[i])
        for a in range(3):
            if full.valid[i, a]:
                nn, npts = int(full.n_nodes[i, a]), int(full.n_pts[i, a])
                assert np.array_equal(sub.nodes[k, a, :nn], full.nodes[i, a, :nn])
                assert np.array_equal(sub.path_param[k, a, :npts], full.path_param[i, a, :npts])
                assert np.array_equal(sub.coeff[k, a, :nn - 1], full.coeff[i, a, :nn - 1])


def test_path_invariants_full_batch(monteblanco, hip_backend):
    """Domain properties at full batch size: consecutive path nodes are joined by lattice edges, the el_length column is
    the offline sample spacing, x/y of node rows interpolate the lattice nodes, node_idx is strictly increasing."""
    lat = monteblanco
    scen, _ = random_scenarios(lat, 1024, seed=21, n_veh=8)
    res = hip_backend.plan_paths(_capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8]))
    checked = 0
    for s in range(0, 1024, 7):
        sl = scen[s]["start_node"][0]
        for a in range(int(res.n_actions[s])):
            if not res.valid[s, a]:
                continue
            nn, npts = int(res.n_nodes[s, a]), int(res.n_pts[s, a])
            nodes, idx = res.nodes[s, a, :nn], res.node_idx[s, a, :nn]
            assert nodes[0] == scen[s]["start_node"][1] and idx[0] == 0 and idx[-1] == npts - 1
            assert np.all(np.diff(idx) > 0)
            pp = res.path_param[s, a, :npts]
            for i in range(nn - 1):
                e = lat.find_edge((sl + i) % lat.num_layers, int(nodes[i]), (sl + i + 1) % lat.num_layers,
                                  int(nodes[i + 1]))
                assert e >= 0
                k0 = lat.samp_ptr[e]
                cnt = idx[i + 1] - idx[i] + (1 if i == nn - 2 else 0)
                assert np.array_equal(pp[idx[i]:idx[i] + cnt, 4], lat.samples[k0:k0 + cnt, 4])
                g = lat.node_pos[lat.layer_off[(sl + i) % lat.num_layers] + nodes[i]]
                assert np.hypot(*(pp[idx[i], 0:2] - g)) < 1e-6
            checked += 1
    assert checked > 100


def test_hip_matches_reference_recordings_on_an_open_track(open_lattice, hip_open, oracle_open):
    """Unclosed track: recordings of the unmodified reference + random scenarios against the oracle, incl. start layers close to
    the end of the track (clamped planning range, reduced-horizon flag) on both kernel forms."""
    recs = load_golden("open_path_calls.npz")
    gen = OnlinePathGenerator(open_lattice, hip_open)
    scen = [replay_path_call(gen, r) for r in recs]
    res = hip_open.plan_paths(_capi.PathsBatch(scen, w_last_edges=recs[0]['w_last_edges']))
    for i, rec in enumerate(recs):
        check_path_output(res.action_sets(i, rec['start_node'][0], open_lattice.num_layers), rec, what="open tick %d" % rec['tick'])
    L = open_lattice.num_layers
    rnd, _ = random_scenarios(open_lattice, 200, seed=77)
    rnd = [s for s in rnd if s['start_node'][0] < L - 1][:128]
    for k, s in enumerate(rnd[:40]):                     # force start layers near the end of the track
        sl = L - 2 - (k % 12)
        s['start_node'] = (sl, int(open_lattice.raceline_index[sl]))
        s['last_nodes'] = None
    for chunk in (rnd, rnd[:9]):
        batch = _capi.PathsBatch(chunk, w_last_edges=[0.0, 0.5, 0.8])
        compare_results(hip_open.plan_paths(batch), oracle_open.plan_paths(batch), open_lattice)
    with pytest.raises(_capi.BackendError):              # the last layer has no planning range
        bad = dict(rnd[0]); bad['start_node'] = (L - 1, 0)
        hip_open.plan_paths(_capi.PathsBatch([bad], w_last_edges=[0.0, 0.5, 0.8]))
