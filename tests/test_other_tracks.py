"""
Further tracks of the reference's inputs/traj_ltpl_cl (all other fixtures are Monteblanco): recordings of the UNMODIFIED reference
(python -m oracle.gen_golden track <name>) on

  zalazone   698 m lap, 66 layers, 3..19 nodes per layer: tight handling course, reduced-horizon 'straight' occurs
  millbrook  429 m lap, 37 layers, 1..14 nodes per layer (layers with ONE node); the 300 m planning range covers 31 of the 37
             layers, so the single opponent is in range on every tick
  lvms      1844 m lap, 159 layers, 9..23 nodes per layer: oval, three race-line followers, all four primitives
  berlin    132 layers, 8..40 nodes per layer (street circuit: the widest layers of all tracks); modena  150 layers, 17..21 nodes per layer.
            Their lattices are not committed (5 MB each): the tests rebuild them with the product's offline build (see lattice_of)

CPU: the oracle at both seams, the planner's host state machine in closed loop, and the offline lattice build against the
lattice exported from the reference's GraphBase. GPU: the same through libltpl_hip.so (first run on an MI355X: round 3, all 13 legs
green, gpurun_out/r03a -> profiles/r03a_other_tracks.txt).
"""
import os

import numpy as np
import pytest

import planner_replay as pr
from helpers import ROOT, load_golden, replay_path_call, check_path_output
from test_oracle_vel_golden import make_vp, replay_vel_call, check_vel_output
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
from graphbasedlocaltrajectoryplanner_amd.path_gen import OnlinePathGenerator

TRACKS = ["zalazone", "millbrook", "lvms", "berlin", "modena"]
BUILT_HERE = ("berlin", "modena")       # no lattice export committed (5 MB each): rebuilt by the product's offline build, see lattice_of
_lattices = {}


def lattice_of(track):
    """Lattice of ``track``: the export of the reference's GraphBase where it is committed; for berlin / modena the lattice the product's
    offline build makes from the race line file -- it reproduces the reference's lattice (topology bit-exact, floats ~1e-13:
    tests/test_offline_build.py against the fingerprints in lattice_digests.json), and the recordings replayed on it below were made by
    the reference on ITS lattice."""
    if track not in _lattices:
        if track in BUILT_HERE:
            from oracle import offline_edges_ref
            from test_offline_build import track as track_arrays
            from graphbasedlocaltrajectoryplanner_amd import offline_build as ob
            _lattices[track] = ob.build_lattice(track_arrays(track), ob.OFFLINE_DEFAULTS, offline_edges_ref.evaluate)
        else:
            _lattices[track] = Lattice.load(os.path.join(ROOT, "tests", "golden", track + "_lattice.npz"))
    return _lattices[track]


def check_paths_against_recordings(lat, backend, track):
    recs = load_golden(track + "_path_calls.npz")
    gen = OnlinePathGenerator(lat, backend)
    assert len(recs) > 30
    for rec in recs:
        sc = replay_path_call(gen, rec)
        res = backend.plan_paths(_capi.PathsBatch([sc], w_last_edges=rec['w_last_edges']))
        check_path_output(res.action_sets(0, rec['start_node'][0], lat.num_layers), rec, what="%s tick %d" % (track, rec['tick']),
                          exact_el=track not in BUILT_HERE)
    return recs


def check_vel_against_recordings(lat, backend, track):
    recs = load_golden(track + "_vel_calls.npz")
    seen = set()
    for i, rec in enumerate(recs):
        out = replay_vel_call(make_vp(backend, lat, rec['state']), rec)
        check_vel_output(out, rec, "%s call %d (%s)" % (track, i, rec['method']))
        seen.add(rec['method'])
    return seen


# ---- CPU ----------------------------------------------------------------------------------------------------------------------
def test_fixtures_differ_from_monteblanco_where_it_matters(monteblanco):
    mz, mm, ml = (lattice_of(t) for t in TRACKS[:3])
    assert int(mm.nodes_in_layer.min()) == 1 and mm.num_layers < monteblanco.max_horizon()[0] + 8      # one-node layers; range ~ lap
    assert ml.num_layers > monteblanco.num_layers and mz.num_layers < monteblanco.num_layers
    assert all(lat.closed for lat in (mz, mm, ml))


@pytest.mark.parametrize("track", TRACKS)
def test_oracle_matches_reference_path_recordings(track):
    from oracle.oracle_lib import OracleBackend
    lat = lattice_of(track)
    recs = check_paths_against_recordings(lat, OracleBackend(lat), track)
    keys = set(k for r in recs for k in r['out']['keys'])
    assert keys >= {"zalazone": {"straight", "follow", "left", "right"}, "millbrook": {"follow"},
                    "lvms": {"straight", "follow", "left", "right"}}.get(track, {"follow", "left", "right"})
    if track == "zalazone":
        assert any(any(r['out']['red_len'].values()) for r in recs)


@pytest.mark.parametrize("track", TRACKS)
def test_oracle_matches_reference_vel_recordings(track):
    from oracle.oracle_lib import OracleBackend
    lat = lattice_of(track)
    seen = check_vel_against_recordings(lat, OracleBackend(lat), track)
    assert 'calc_vel_profile_follow' in seen and ('calc_vel_profile' in seen or track == "millbrook")        # millbrook: follow only


@pytest.mark.parametrize("track,must_see", [("zalazone", {"straight", "follow", "left", "right"}), ("millbrook", {"follow"}),
                                            ("lvms", {"straight", "follow", "left", "right"}),
                                            ("berlin", {"follow", "left", "right"}), ("modena", {"follow", "left", "right"})])
def test_host_state_machine_in_closed_loop(track, must_see):
    from oracle.planner_host import HostPlannerBackend
    lat = lattice_of(track)
    ticks = pr.load_ticks(track)
    seen = pr.replay(HostPlannerBackend(lat).planner(1), lat, ticks)
    assert len(ticks) == 900 and must_see <= seen['keys'] and seen['full'] >= 15, seen


@pytest.mark.parametrize("track", TRACKS[:3])
def test_offline_build_reproduces_the_reference_lattice(track):
    from oracle import offline_edges_ref
    from test_offline_build import track as track_arrays, check_against_reference_export
    from graphbasedlocaltrajectoryplanner_amd import offline_build as ob
    check_against_reference_export(ob.build_lattice(track_arrays(track), ob.OFFLINE_DEFAULTS, offline_edges_ref.evaluate), track)


# ---- GPU ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip_of():
    made = {}

    def get(track):
        if track not in made:
            made[track] = _capi.HipBackend(lattice_of(track))
        return made[track]
    return get


@pytest.mark.gpu
@pytest.mark.parametrize("track", TRACKS)
def test_hip_matches_reference_recordings_and_the_oracle(hip_of, track):
    from oracle.oracle_lib import OracleBackend
    from test_gpu_paths import compare_results
    from scenarios import random_scenarios
    lat, hip = lattice_of(track), hip_of(track)
    check_paths_against_recordings(lat, hip, track)
    check_vel_against_recordings(lat, hip, track)
    scen, _ = random_scenarios(lat, 300, seed=5)
    for nb in (300, 7):                                       # batch kernel and the four-wave form
        batch = _capi.PathsBatch(scen[:nb], w_last_edges=[0.0, 0.5, 0.8])
        compare_results(hip.plan_paths(batch), OracleBackend(lat).plan_paths(batch), lat)


@pytest.mark.gpu
@pytest.mark.parametrize("track", TRACKS)
def test_planner_closed_loop_on_the_device(hip_of, track):
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    lat = lattice_of(track)
    planner = Planner(hip_of(track), 1)
    seen = pr.replay(planner, lat, pr.load_ticks(track))
    assert seen['full'] >= 15
    planner.close()


@pytest.mark.gpu
@pytest.mark.parametrize("track", TRACKS[:3])
def test_device_build_reproduces_the_reference_lattice(hip_of, track):
    from test_offline_build import track as track_arrays, check_against_reference_export
    from graphbasedlocaltrajectoryplanner_amd import offline_build as ob
    lat = ob.build_lattice(track_arrays(track), ob.OFFLINE_DEFAULTS, ob.edges_on_device(hip_of(track).lib))
    check_against_reference_export(lat, track)
