"""
GPU: the FLEET (ltpl_fleet_*, include/ltpl_hip.h ABI v5) -- planners whose iterative memory lives in device memory and is advanced by
kernels (one wave64 per planner, csrc/fleet_core.hpp) around the path kernel and the velocity kernel -- replayed in closed loop against the
tick recordings of the unmodified reference, through the per-call entry points and through the tape (many ticks back to back without host
synchronisation), and compared with the host planner (ltpl_planner_*) on the same device arithmetic.
"""
import numpy as np
import pytest

import planner_replay as pr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(monteblanco):
    from graphbasedlocaltrajectoryplanner_amd import _capi
    return _capi.HipBackend(monteblanco)


@pytest.mark.parametrize("name,n,must_see", [
    ("c2", 1, {"straight", "follow", "left", "right"}),
    ("c1", 1, {"straight", "follow"}),
    ("zonewall", 3, {"straight", "follow", "right"}),
    ("ggdrop", 1, {"straight"}),                              # backup branch: second velocity launch
    ("overtake", 70, {"follow", "left", "right", "emergency"}),   # >= 64 planners: one-wave batch path kernel; emergency: third launch
    ("ggmap", 2, {"follow", "emergency"}),                    # location dependent friction: rows per job (k_vel_profile GG, SEL 2 / 3 / 1)
    ("ggmapdrop", 2, {"straight", "emergency"}),              # ... losing grip: backup brake jobs on the backup path's own rows (JB with rows)
    ("car2ggmap", 2, {"follow", "right", "emergency"}),       # the other car on the friction map
    ("car2ggdrop", 1, {"straight", "emergency"}),
])
def test_closed_loop_replay_matches_reference_recordings(hip, monteblanco, name, n, must_see):
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    ticks = pr.load_ticks(name)
    fleet = Fleet(hip, n)
    seen = pr.replay(fleet, monteblanco, ticks, scen=n - 1)
    assert must_see <= seen['keys'] and seen['full'] >= 15, seen
    fleet.close()


def test_velocity_smoothing_window(hip, monteblanco):
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    fleet = Fleet(hip, 1, filt_window_width=5)
    seen = pr.replay(fleet, monteblanco, pr.load_ticks("filt5"))
    assert seen['full'] >= 15
    fleet.close()


def test_open_track(open_lattice):
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    hip = _capi.HipBackend(open_lattice)
    fleet = Fleet(hip, 2)
    seen = pr.replay(fleet, open_lattice, pr.load_ticks("open"), scen=1)
    assert {"straight", "follow", "right"} <= seen['keys'] and seen['full'] >= 15
    fleet.close()


@pytest.mark.parametrize("track", ["millbrook", "berlin", "lvms", "modena", "zalazone"])
def test_other_tracks(track):
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    from test_other_tracks import lattice_of
    lat = lattice_of(track)
    fleet = Fleet(_capi.HipBackend(lat), 1)
    seen = pr.replay(fleet, lat, pr.load_ticks(track))
    assert seen['full'] >= 15
    fleet.close()


def group_inputs(lat, t):
    va = t['vel_args']
    return dict(prev_action=t['action_id_sel'], t_now=t['t'], vehicles=pr.vehicles_of_tick(t), zone_gids=pr.zone_gids_of_tick(lat, t),
                pos_est=t['pos_est'], vel_est=va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])


def test_tape_of_mixed_recordings(hip, monteblanco):
    """192 planners in four groups, each group replaying another recording (c2 / overtake / zonewall / ggdrop: all primitives, dropped
    overtakes, emergency profiles, the backup branch), 300 ticks back to back from pre-uploaded inputs without any host synchronisation;
    then planners of every group against the recording's tick 299, and against the host planner's state after the same ticks."""
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    names, per, T = ("c2", "overtake", "zonewall", "ggdrop"), 48, 300
    recs = [pr.load_ticks(nm) for nm in names]
    fleet = Fleet(hip, per * len(names))
    for g, ticks in enumerate(recs):
        st = ticks[0]['start']
        for p in (g * per, g * per + 1, g * per + per - 1):           # the planners looked at below (+ one that is not: see the skip)
            fleet.set_start(p, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    # planners without a start pose would stop the run with their error: give all the same pose of their group
    for g, ticks in enumerate(recs):
        st = ticks[0]['start']
        for p in range(g * per + 2, g * per + per - 1):
            fleet.set_start(p, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    for k in range(T):
        fleet.tape_append_groups([(per, group_inputs(monteblanco, ticks[k])) for ticks in recs], ax_max_machines=recs[0][k]['vel_args']['ax_max_machines'])
    ms = fleet.tape_run(0, T)
    assert ms > 0.0
    for g, ticks in enumerate(recs):
        t = ticks[T - 1]
        for p in (g * per, g * per + per - 1):
            traj, ids, ref = fleet.trajectories(p)
            pr.check_trajectories(traj, ids, ref, t, "%s tick %d planner %d" % (names[g], t['tick'], p))
    # the host planner on the same device arithmetic, same ticks: identical state
    host = Planner(hip, 1)
    ticks = recs[1]
    st = ticks[0]['start']
    host.set_start(0, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    for t in ticks[:T]:
        va = t['vel_args']
        host.calc_paths([t['action_id_sel']], [t['t']], [pr.vehicles_of_tick(t)], [pr.zone_gids_of_tick(monteblanco, t)])
        host.calc_vel_profile([t['pos_est']], va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                              ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])
    ta, tb = fleet.trajectories(per + 5), host.trajectories(0)
    assert list(ta[0].keys()) == list(tb[0].keys()) and ta[1] == tb[1] and ta[2]['cut_index_pos'] == tb[2]['cut_index_pos']
    for k in ta[0]:
        # (not bit-equal: the fleet solves its forward-backward jobs one lane per job on fp32 (|kappa|, element length) operands like the
        # batch velocity stage, the host planner goes through the wave-per-job fp64 kernel; 1e-5 relative is the contract)
        pr.check_traj(ta[0][k][0], tb[0][k][0], "fleet vs host planner / %s" % k)
        assert np.max(np.abs(ta[0][k][0][:, 5] - tb[0][k][0][:, 5])) <= 2e-6 * max(1.0, float(np.max(np.abs(tb[0][k][0][:, 5])))), k
    pa, pb = fleet.paths(per + 5), host.paths(0)
    assert pa['keys'] == pb['keys'] and pa['nodes'] == pb['nodes'] and pa['node_idx'] == pb['node_idx']
    host.close()
    fleet.close()


def test_a_failing_planner_keeps_its_error_and_does_not_disturb_the_others(hip, monteblanco):
    from graphbasedlocaltrajectoryplanner_amd._capi import BackendError
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    ticks = pr.load_ticks("c1")
    fleet = Fleet(hip, 2)
    st = ticks[0]['start']
    fleet.set_start(1, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    for t in ticks[:60]:
        veh, zg, va = pr.vehicles_of_tick(t), pr.zone_gids_of_tick(monteblanco, t), t['vel_args']
        with pytest.raises(BackendError, match="planner 0: no start node"):
            fleet.calc_paths([t['action_id_sel']] * 2, [t['t']] * 2, [veh] * 2, [zg] * 2)
        with pytest.raises(BackendError, match="planner 0"):
            fleet.calc_vel_profile([t['pos_est']] * 2, va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                                   ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])
        traj, ids, ref = fleet.trajectories(1)
        pr.check_trajectories(traj, ids, ref, t, "tick %d" % t['tick'])
    fleet.close()


def test_split_entry_points_equal_the_fused_ones(hip, monteblanco):
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    from test_fleet_host_logic import _SplitCalls
    fleet = Fleet(hip, 2)
    seen = pr.replay(_SplitCalls(fleet), monteblanco, pr.load_ticks("zonewall"), scen=1)
    assert seen['full'] >= 15
    fleet.close()


def test_a_fleet_of_different_cars_on_the_device(hip, monteblanco):
    """ABI v6 on the MI355X: per-planner vel_max / machine tables in one ltpl_fleet_calc_vel_profile call per tick -- 2 x 33 planners replay
    the reference's c2 (default car) and car2 (vel_max 42 m/s, 18-row machine table) recordings side by side; lane-per-job forward-backward
    jobs and wave-per-job follow jobs both take the car from the job."""
    from test_fleet_host_logic import two_cars_replay
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    fleet = Fleet(hip, 66)
    seen = two_cars_replay(fleet, monteblanco, 33, 300, check_every=7)
    fleet.close()
    assert {"follow", "right", "emergency"} <= seen


def test_friction_rows_next_to_constant_tuples_on_the_device(hip, monteblanco):
    """local_gg per vehicle on the MI355X: 2 x 33 planners of one fleet, the first half with the friction MAP of the 'ggmap' recording
    (rows per path coordinate, OTH.py:649-666), the second half with a constant tuple on 'c2' -- forward-backward jobs with rows go to the
    wave-per-job kernel (SEL 3), those with constants stay on the lane kernel, in the same call."""
    from test_fleet_host_logic import friction_rows_next_to_constants
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    fleet = Fleet(hip, 66)
    keys = friction_rows_next_to_constants(fleet, monteblanco, n_ticks=200, reps=33)
    fleet.close()
    assert {"follow", "emergency", "right"} <= keys, keys


def test_friction_rows_that_do_not_match_the_path_on_the_device(hip, monteblanco):
    from graphbasedlocaltrajectoryplanner_amd._capi import BackendError
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    ticks = pr.load_ticks("c1")
    fleet = Fleet(hip, 2)
    st, t = ticks[0]['start'], ticks[0]
    for p in (0, 1):
        fleet.set_start(p, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    veh, zg, va = pr.vehicles_of_tick(t), pr.zone_gids_of_tick(monteblanco, t), t['vel_args']
    fleet.calc_paths([t['action_id_sel']] * 2, [t['t']] * 2, [veh] * 2, [zg] * 2)
    short = {k: [np.full((v.shape[0] - 1, 2), 5.0)] for k, v in fleet.paths(0)['path_param'].items()}
    with pytest.raises(BackendError, match="planner 0: local_gg rows"):
        fleet.calc_vel_profile([t['pos_est']] * 2, va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=[short, tuple(va['local_gg'])],
                               ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])
    traj, ids, ref = fleet.trajectories(1)
    pr.check_trajectories(traj, ids, ref, t, "the neighbour of the failing planner")
    fleet.close()


def test_fused_tape_kernels_equal_the_separate_ones(hip, monteblanco, monkeypatch):
    """Tape runs execute paths_post + vel_a and vel_c | vel_d + the next tick's paths_pre as one kernel each (round 4). Same stages, same
    order per planner: the state after 151 ticks of four recordings (with and without emergency launches, i.e. both tail variants) must
    be IDENTICAL, array for array, to the run with one kernel per stage (LTPL_FLEET_NO_FUSE=1, read when the fleet is created), also when
    the tape is run in several segments."""
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    names, per, T = ("c2", "overtake", "zonewall", "ggdrop"), 8, 151     # (the overtake recording asks for the emergency profile every third tick: 150 is one)
    recs = [pr.load_ticks(nm) for nm in names]

    def run(segments):
        fleet = Fleet(hip, per * len(names))
        for g, ticks in enumerate(recs):
            st = ticks[0]['start']
            fleet.set_start_range(g * per, (g + 1) * per, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
        for k in range(T):
            fleet.tape_append_groups([(per, group_inputs(monteblanco, ticks[k])) for ticks in recs], ax_max_machines=recs[0][k]['vel_args']['ax_max_machines'])
        for a, b in segments:
            fleet.tape_run(a, b - a)
        out = [(fleet.trajectories(p), fleet.paths(p)) for p in (0, per, 2 * per + 3, 4 * per - 1)]
        fleet.close()
        return out
    fused = run([(0, T)])
    fused_seg = run([(0, 1), (1, 70), (70, T)])
    monkeypatch.setenv("LTPL_FLEET_NO_FUSE", "1")
    plain = run([(0, T)])
    for other in (fused_seg, plain):
        for (ta, pa), (tb, pb) in zip(fused, other):
            assert list(ta[0].keys()) == list(tb[0].keys()) and ta[1] == tb[1] and ta[2]['cut_index_pos'] == tb[2]['cut_index_pos']
            for k in ta[0]:
                assert np.array_equal(ta[0][k][0], tb[0][k][0]), k
            assert pa['keys'] == pb['keys'] and pa['nodes'] == pb['nodes']
            assert all(np.array_equal(pa['path_param'][k], pb['path_param'][k]) for k in pa['keys'])
    assert "emergency" in fused[1][0][0]                       # the overtake group ran with the emergency launch (tail variant D)


def test_emergency_profile_on_a_backup_plan_with_friction_rows_on_the_device(hip, monteblanco):
    """The device's detection of the situation in which the reference raises "Length of loc_gg and kappa must be equal!" (see
    tests/test_fleet_host_logic.py); the neighbour planner, which does not ask for the emergency profile, is served."""
    from graphbasedlocaltrajectoryplanner_amd._capi import BackendError
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    ticks = pr.load_ticks("ggmapdrop")
    fleet = Fleet(hip, 2)
    pr.replay(fleet, monteblanco, ticks, n_ticks=300, scen=1)
    t = ticks[300]
    va = t['vel_args']
    fleet.calc_paths([t['action_id_sel']] * 2, [t['t']] * 2, [pr.vehicles_of_tick(t)] * 2, [pr.zone_gids_of_tick(monteblanco, t)] * 2)
    lgg = pr.local_gg_of_tick(t, fleet.paths(0)['path_param'])
    with pytest.raises(BackendError, match="planner 0: emergency profile.*Length of loc_gg and kappa must be equal"):
        fleet.calc_vel_profile([t['pos_est']] * 2, va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=[lgg, lgg],
                               ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'], incl_emerg_traj=[True, False])
    traj, ids, ref = fleet.trajectories(1)
    pr.check_trajectories(traj, ids, ref, t, "the neighbour of the failing planner")
    fleet.close()


def test_machine_tables_and_friction_rows_per_planner_on_the_device(hip, monteblanco):
    """2 x 33 planners: the other car on the friction map (machine table 0, rows) next to the default car with a constant tuple (table 1) in
    the same ltpl_fleet_calc_vel_profile calls -- the wave-per-job kernels in their rows form take the job's own table, the lane kernel
    stages both tables and leaves the jobs with rows to them."""
    from test_fleet_host_logic import cars_and_rows_replay
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    fleet = Fleet(hip, 66)
    seen = cars_and_rows_replay(fleet, monteblanco, 33, 160, check_every=5)
    fleet.close()
    assert {"follow", "right", "emergency"} <= seen, seen


def test_a_tape_with_machine_tables_and_friction_rows(hip, monteblanco):
    """The TAPE with per-planner tables and friction rows (fused stage kernels, rows form of the velocity launches): the packed inputs of a
    per-call run of 120 ticks are appended to the tape of a second fleet, which must end in the identical state."""
    from test_fleet_host_logic import cars_and_rows_replay
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    T, reps = 120, 3
    a, b = Fleet(hip, 2 * reps), Fleet(hip, 2 * reps)
    ticks = []
    cars_and_rows_replay(a, monteblanco, reps, T, check_every=20, collect=ticks)
    recs = [pr.load_ticks("car2ggmap"), pr.load_ticks("c2")]
    for g, rec in enumerate(recs):
        st = rec[0]['start']
        b.set_start_range(g * reps, (g + 1) * reps, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    for pi, vi, _k0, _k1 in ticks:
        b.tape_append_packed(pi, vi)
    assert b.tape_run(0, T) > 0.0
    for q in (0, reps - 1, reps, 2 * reps - 1):
        (ta, ia, ra), (tb, ib, rb) = a.trajectories(q), b.trajectories(q)
        assert list(ta.keys()) == list(tb.keys()) and ia == ib and ra['cut_index_pos'] == rb['cut_index_pos']
        for k in ta:
            assert np.array_equal(ta[k][0], tb[k][0]), (q, k)
        pr.check_trajectories(tb, ib, rb, recs[0 if q < reps else 1][T - 1], "tape planner %d" % q)
    a.close(); b.close()


# ---- the LANE form of the follow jobs (k_fleet_follow_lanes) at test sizes ------------------------------------------------------------
# The lane-per-job follow kernel is the default from 12 288 planners on (fleet_dev.hpp) -- the size of the bench's fleet legs -- and no test
# creates a fleet that large: LTPL_FLEET_FOLLOW_WAVES=0 (read when the fleet is created) forces it for the small fleets below, so the kernel
# the headline fleet numbers run on is held to the recordings of the reference and to the wave-per-job form (advisor finding, round 5).

@pytest.mark.parametrize("name,n,must_see", [
    ("c2", 3, {"straight", "follow", "left", "right"}),
    ("overtake", 70, {"follow", "left", "right", "emergency"}),
    ("zonewall", 3, {"straight", "follow", "right"}),
    ("car2", 2, {"follow"}),
])
def test_lane_form_of_the_follow_jobs_replays_the_recordings(hip, monteblanco, monkeypatch, name, n, must_see):
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    monkeypatch.setenv("LTPL_FLEET_FOLLOW_WAVES", "0")
    ticks = pr.load_ticks(name)
    fleet = Fleet(hip, n)
    seen = pr.replay(fleet, monteblanco, ticks, scen=n - 1)
    assert must_see <= seen['keys'] and seen['full'] >= 15, seen
    fleet.close()


def test_lane_form_next_to_jobs_with_friction_rows_in_one_call(hip, monteblanco, monkeypatch):
    """One job table, both kernels: the follow jobs of the planners on the friction MAP (rows per path coordinate) stay wave-per-job, those
    of the planners with a constant tuple go to the lane kernel -- in the same ltpl_fleet_calc_vel_profile call, every tick, against the
    recordings of both halves."""
    from test_fleet_host_logic import friction_rows_next_to_constants
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    monkeypatch.setenv("LTPL_FLEET_FOLLOW_WAVES", "0")
    fleet = Fleet(hip, 66)
    keys = friction_rows_next_to_constants(fleet, monteblanco, n_ticks=200, reps=33)
    fleet.close()
    assert {"follow", "emergency", "right"} <= keys, keys


def test_lane_form_and_wave_form_of_the_follow_jobs_agree(hip, monteblanco, monkeypatch):
    """The same mixed tape (four recordings side by side, 120 ticks, state carried on the device) through a fleet with the lane form and one
    with the wave-per-job form of the follow jobs: every looked-at planner ends with the same keys / ids / cut indices / node lists and
    trajectories that agree to 1e-9 relative (both forms are fp64 since round 6; they order the arithmetic differently -- w = v^2 recurrence
    per lane against the systolic wave -- so not bit for bit)."""
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    names, per, T = ("c2", "overtake", "zonewall", "c1"), 18, 120
    recs = [pr.load_ticks(nm) for nm in names]

    def run(form):
        monkeypatch.setenv("LTPL_FLEET_FOLLOW_WAVES", form)
        fleet = Fleet(hip, per * len(names))
        for g, ticks in enumerate(recs):
            st = ticks[0]['start']
            fleet.set_start_range(g * per, (g + 1) * per, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
        for k in range(T):
            fleet.tape_append_groups([(per, group_inputs(monteblanco, ticks[k])) for ticks in recs], ax_max_machines=recs[0][k]['vel_args']['ax_max_machines'])
        fleet.tape_run(0, T)
        out = [(fleet.trajectories(p), fleet.paths(p)) for g in range(len(names)) for p in (g * per, g * per + per // 2, g * per + per - 1)]
        dig = fleet.digest()
        fleet.close()
        return out, dig
    (lanes, dl), (waves, dw) = run("0"), run("1")
    n_follow = 0
    for (ta, pa), (tb, pb) in zip(lanes, waves):
        assert list(ta[0].keys()) == list(tb[0].keys()) and ta[1] == tb[1] and ta[2]['cut_index_pos'] == tb[2]['cut_index_pos'] and ta[2]['cut_layer'] == tb[2]['cut_layer']
        assert pa['keys'] == pb['keys'] and pa['nodes'] == pb['nodes'] and pa['node_idx'] == pb['node_idx']
        n_follow += int("follow" in ta[0])
        for k in ta[0]:
            x, y = ta[0][k][0], tb[0][k][0]
            assert x.shape == y.shape, k
            assert np.array_equal(x[:, :5], y[:, :5]), k                                   # s, x, y, psi, kappa: the same path kernel
            assert float(np.max(np.abs(x[:, 5] - y[:, 5]))) <= 1e-9 * max(1.0, float(np.max(np.abs(y[:, 5])))), (k, "vx")
            assert float(np.max(np.abs(x[:, 6] - y[:, 6]))) <= 1e-7 * max(5.0, float(np.max(np.abs(y[:, 5]))) ** 2 / 2.0), (k, "ax")
    assert n_follow >= 3
    # ... and every planner of both fleets through the device digest (integers exact, floats 1e-9)
    assert dl.shape == dw.shape and np.array_equal(dl[:, :5], dw[:, :5])
    assert np.allclose(dl, dw, rtol=1e-9, atol=1e-9)
