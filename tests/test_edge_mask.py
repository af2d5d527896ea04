"""
SURVEY.md section 8a row M2 compared DIRECTLY: the obstacle x edge mask (GraphBase.get_intersec_edges_in_range, GraphBase.py:567-646;
the edges gen_local_node_template.py:164-203 deletes from the "default" filter), not only through the paths it shapes.

  reference  tests/golden/fresh_ticks.npz carries, per scenario, every edge the unmodified reference's get_intersec_edges_in_range
             returned while main_online_path_gen ran (oracle/gen_golden_fresh.py)
  oracle     oracle_plan_paths_mask: the restatement's blocked-edge array                     (CPU: equals the reference's set)
  HIP        ltpl_plan_paths_mask: the LDS bitmap of the path kernel's phase 2 (capsule cull: certain MISS / certain HIT from the
             per-edge capsule table, exact fp64 sample test in between), exported by the kernel itself, one-wave batch form and
             four-wave latency form                              (GPU: equals the oracle's array bit for bit on active edges, and the
                                                                 reference's sets)

The kernel tests every edge of a window inside the planning range; the reference / oracle only edges whose end nodes the zone filter
left active (an edge at a removed node is unusable either way). The comparison is therefore made on the edges with two active end
nodes -- a false HIT or a false MISS of the cull on ANY such edge fails, whether or not the edge lies on an optimum.
"""
import numpy as np
import pytest

from helpers import load_golden, replay_path_call
from scenarios import random_scenarios
from test_fresh_tick_golden import records, scenario_of
from test_gpu_paths import compare_results
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.path_gen import OnlinePathGenerator


def active_edges(lat, batch):
    """bool [n_scen, E]: both end nodes of the edge are not removed by the scenario's zone filter."""
    dst = lat.edge_dst_gid()
    sl, sn, _, _ = lat.edge_endpoints()
    src = lat.layer_off[sl] + sn
    act = np.ones((batch.n_scen, lat.num_edges), bool)
    for s in range(batch.n_scen):
        z = np.asarray(batch.zone_gid[batch.zone_off[s]:batch.zone_off[s + 1]], dtype=np.int64)
        if z.size:
            removed = np.zeros(lat.num_nodes, bool)
            removed[z] = True
            act[s] = ~(removed[src] | removed[dst])
    return act


def reference_mask(lat, recs):
    m = np.zeros((len(recs), lat.num_edges), np.uint8)
    for i, r in enumerate(recs):
        for a, b, c, d in np.asarray(r["blocked_edges"]).reshape(-1, 4):
            e = lat.find_edge(int(a), int(b), int(c), int(d))
            assert e >= 0
            m[i, e] = 1
    return m


def fresh_batch(lat):
    recs = records()
    return recs, _capi.PathsBatch([scenario_of(r) for r in recs], w_last_edges=recs[0]["scen"]["w_last_edges"])


def test_oracle_mask_equals_the_reference_sets(monteblanco, oracle_backend):
    recs, batch = fresh_batch(monteblanco)
    _, got = oracle_backend.plan_paths_mask(batch)
    ref = reference_mask(monteblanco, recs)
    assert np.array_equal(got, ref)
    assert int(ref.sum()) > 2000 and int((ref.sum(axis=1) > 0).sum()) > 80       # the fixture does block edges


def hip_vs_oracle(lat, hip, orc, batch, forms=(1, 4), min_blocked=1):
    ores, omask = orc.plan_paths_mask(batch)
    act = active_edges(lat, batch)
    assert int(omask.sum()) >= min_blocked
    for nw in forms:
        res, mask = hip.plan_paths_mask(batch, team_waves=nw)
        diff = (mask != omask) & act
        assert not diff.any(), "team of %d waves: %d edges differ (first: scenario %d edge %d, hip %d oracle %d)" % (
            nw, int(diff.sum()), *[int(x[0]) for x in np.nonzero(diff)], int(mask[diff][0]), int(omask[diff][0]))
        compare_results(res, ores, lat)                         # the diagnostic call plans like ltpl_plan_paths
    return omask


@pytest.mark.gpu
def test_hip_mask_equals_reference_and_oracle_on_the_fresh_scenarios(monteblanco, hip_backend, oracle_backend):
    recs, batch = fresh_batch(monteblanco)
    hip_vs_oracle(monteblanco, hip_backend, oracle_backend, batch, min_blocked=2000)
    ref = reference_mask(monteblanco, recs)
    act = active_edges(monteblanco, batch)
    for nw in (1, 4):
        _, mask = hip_backend.plan_paths_mask(batch, team_waves=nw)
        assert np.array_equal(mask * act, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["c2_path_calls.npz", "c1_path_calls.npz", "zonewall_path_calls.npz"])
def test_hip_mask_equals_oracle_on_the_recorded_calls(monteblanco, hip_backend, oracle_backend, fixture):
    recs = load_golden(fixture)
    gen = OnlinePathGenerator(monteblanco, hip_backend)
    batch = _capi.PathsBatch([replay_path_call(gen, r) for r in recs], w_last_edges=recs[0]['w_last_edges'])
    hip_vs_oracle(monteblanco, hip_backend, oracle_backend, batch, min_blocked=100)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_veh", [(21, 8), (22, 16), (23, 96)])
def test_hip_mask_equals_oracle_on_random_scenarios(monteblanco, hip_backend, oracle_backend, seed, n_veh):
    """n_veh = 96 with one prediction each = 192 obstacle positions per scenario (the capacity limit)."""
    scen, _ = random_scenarios(monteblanco, 96 if n_veh == 96 else 256, seed=seed, n_veh=n_veh)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    hip_vs_oracle(monteblanco, hip_backend, oracle_backend, batch, min_blocked=1000)


@pytest.mark.gpu
def test_hip_mask_equals_oracle_on_c3():
    from oracle.oracle_lib import OracleBackend
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice, scattered_obstacle_scenarios
    lat = c3_lattice()
    hip, orc = _capi.HipBackend(lat), OracleBackend(lat)
    scen, _ = scattered_obstacle_scenarios(lat, 128, n_obj=32, seed=4)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    hip_vs_oracle(lat, hip, orc, batch, min_blocked=1000)
    hip.close()
