"""GPU: seam (2) (velocity kernels) and the fused tick through the C ABI against recordings and the oracle."""
import numpy as np
import pytest

from helpers import load_golden, assert_close_rel, assert_vx_elementwise, assert_ax_elementwise
from scenarios import random_scenarios, raceline_state
from test_oracle_vel_golden import make_vp, replay_vel_call, check_vel_output
from graphbasedlocaltrajectoryplanner_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture", ["c2_vel_calls.npz", "c1_vel_calls.npz", "zonewall_vel_calls.npz"])
def test_hip_matches_reference_vel_recordings(monteblanco, hip_backend, fixture):
    recs = load_golden(fixture)
    seen = set()
    for i, rec in enumerate(recs):
        vp = make_vp(hip_backend, monteblanco, rec['state'])
        out = replay_vel_call(vp, rec)
        check_vel_output(out, rec, "%s call %d (%s)" % (fixture, i, rec['method']))
        seen.add(rec['method'])
    assert {'calc_vel_profile', 'calc_vel_profile_follow'} <= seen or fixture != "c2_vel_calls.npz"


def random_jobs(lat, rng, n_jobs, varying_gg):
    jobs = []
    track_len = float(lat.glob_rl[-1, 0])
    for _ in range(n_jobs):
        n = int(rng.integers(2, 400))
        mode = int(rng.integers(0, 3))
        # curvature profile: piecewise smooth with straights (exact zeros) and tight corners
        kappa = 0.08 * np.sin(np.linspace(0, rng.uniform(1, 12), n) + rng.uniform(0, 6)) * rng.uniform(0, 1)
        kappa[np.abs(kappa) < 0.004] = 0.0
        el = rng.uniform(1.5, 3.5, n - 1)
        if varying_gg:
            gg = np.column_stack((rng.uniform(3.0, 8.0, n), rng.uniform(3.0, 8.0, n)))
        else:
            gg = np.ones((n, 2)) * rng.uniform(3.0, 9.0, 2)
        job = {"mode": mode, "kappa": kappa, "loc_gg": gg, "v_start": float(rng.uniform(0, 70))}
        if mode == _capi.VEL_FB:
            job["el_lengths"] = el
            job["v_end"] = float(rng.uniform(0, 60)) if rng.random() < 0.8 else None
        elif mode == _capi.VEL_BRAKE:
            job["el_lengths"] = el
        else:
            job["el_lengths"] = np.append(el, 0.0)
            x, y, _, v = raceline_state(lat, rng.uniform(0, track_len))
            job.update(v_ego=job["v_start"] + rng.uniform(-1, 1), v_obj=float(v) * rng.uniform(0.1, 1.0),
                       safety_d=float(rng.uniform(5, 40)), obj_dist=float(rng.uniform(-5, 400)),
                       obj_pos=(float(x + rng.uniform(-2, 2)), float(y + rng.uniform(-2, 2))))
        jobs.append(job)
    return jobs


@pytest.mark.parametrize("exp,axm,ctrl,varying_gg", [
    (1.0, [[100.0, 5.0]], "PD", False),
    (1.0, [[0.0, 6.0], [36.0, 6.0], [48.0, 4.8], [60.0, 3.9], [72.0, 2.5]], "PD", True),
    (2.0, [[100.0, 5.0]], "PDtan", False),
    (1.5, [[0.0, 6.0], [72.0, 2.5]], "PD", True),
])
def test_hip_vel_matches_oracle_on_random_jobs(monteblanco, hip_backend, oracle_backend, exp, axm, ctrl, varying_gg):
    rng = np.random.default_rng(int(exp * 10) + len(axm))
    params = _capi.VelParamSet(dyn_model_exp=exp, drag_coeff=0.85, m_veh=1000.0, len_veh=monteblanco.veh_length,
                               v_max=float(rng.uniform(40, 100)), ax_max_machines=axm, follow_control_type=ctrl,
                               follow_control_params={"c_p": 1.15, "k_d": 0.025, "k_p": 0.2, "tan_w": 15.0})
    jobs = random_jobs(monteblanco, rng, 300, varying_gg)
    got = hip_backend.vel_profile(params, jobs)
    exp_ = oracle_backend.vel_profile(params, jobs)
    n_flag_mismatch = 0
    for i, ((vx, tc, vb), (rx, rtc, rvb)) in enumerate(zip(got, exp_)):
        assert_close_rel(vx, rx, what="job %d mode %d n %d" % (i, jobs[i]["mode"], jobs[i]["kappa"].size))
        assert_vx_elementwise(vx, rx, "job %d mode %d" % (i, jobs[i]["mode"]))
        assert tc == rtc
        n_flag_mismatch += int(vb != rvb)
    assert n_flag_mismatch == 0


def make_tick_inputs(lat, n, seed, n_veh=8):
    scen, vels = random_scenarios(lat, n, seed=seed, n_veh=n_veh)
    rng = np.random.default_rng(seed + 1000)
    params = _capi.VelParamSet(len_veh=lat.veh_length)
    vplan = rng.uniform(0.0, 60.0, n)
    vplan[::17] = 0.0
    pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    pos = pos + rng.uniform(-0.3, 0.3, pos.shape)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    vel = _capi.TickVelBatch(params, n, vplan, vplan + rng.uniform(-1, 1, n), pos,
                             np.concatenate(vels) if n_veh else np.zeros(0))
    return batch, vel


def compare_tick(res, vres, ref, vref):
    from test_gpu_paths import compare_results
    compare_results(res, ref, None)
    assert np.array_equal(vres.too_close, vref.too_close)
    assert np.array_equal(vres.vel_bound, vref.vel_bound)
    for s in range(res.n_scen):
        for a in range(int(res.n_actions[s])):
            if res.valid[s, a]:
                n = int(res.n_pts[s, a])
                assert_close_rel(vres.vx[s, a, :n], vref.vx[s, a, :n], what="vx s%d a%d" % (s, a))
                # ax = d(v^2) / (2 ds): compare against the scale of v^2 / ds
                scale = max(float(np.max(np.abs(vref.vx[s, a, :n]))) ** 2 / 2.0, 5.0)
                err = float(np.max(np.abs(vres.ax[s, a, :n] - vref.ax[s, a, :n])))
                assert err <= 1e-5 * scale, "ax s%d a%d err %.3e" % (s, a, err)
                # ... and sample by sample (north_star: velocity profiles within 1e-5 relative; every operand of the stage is fp64)
                assert_vx_elementwise(vres.vx[s, a, :n], vref.vx[s, a, :n], "s%d a%d" % (s, a))
                assert_ax_elementwise(vres.ax[s, a, :n], vref.ax[s, a, :n], "s%d a%d" % (s, a))


@pytest.mark.parametrize("seed,n_veh", [(0, 8), (7, 3), (9, 0)])
def test_fused_tick_matches_oracle(monteblanco, hip_backend, oracle_backend, seed, n_veh):
    batch, vel = make_tick_inputs(monteblanco, 256, seed, n_veh)
    res, vres = hip_backend.tick_batch(batch, vel)
    ref, vref = oracle_backend.tick_batch(batch, vel)
    compare_tick(res, vres, ref, vref)
    if n_veh >= 3:
        assert int(vres.too_close.sum()) > 0 or int((res.action_id == _capi.ACT_FOLLOW).sum()) > 0


def test_resident_batch_equals_tick_batch(monteblanco, hip_backend):
    batch, vel = make_tick_inputs(monteblanco, 512, 3, 8)
    res, vres = hip_backend.tick_batch(batch, vel)
    hip_backend.batch_upload(batch, vel)
    ms = hip_backend.batch_run(reps=3, timed=True)
    assert ms > 0.0
    res2, vres2 = hip_backend.batch_download()
    for name in ("valid", "action_id", "n_pts", "n_nodes", "reduced"):
        assert np.array_equal(getattr(res, name), getattr(res2, name)), name
    assert np.array_equal(vres.vel_bound, vres2.vel_bound)
    # (entries of the capacity slabs behind n_nodes / n_pts are unspecified -- include/ltpl_hip.h -- and the resident pipeline rotates through
    # three buffer sets: compare what is defined, bit for bit)
    for s, a in zip(*np.nonzero(res.valid)):
        nn, npt = int(res.n_nodes[s, a]), int(res.n_pts[s, a])
        assert np.array_equal(res.nodes[s, a, :nn], res2.nodes[s, a, :nn]) and np.array_equal(res.node_idx[s, a, :nn], res2.node_idx[s, a, :nn])
        assert np.array_equal(res.coeff[s, a, :nn - 1], res2.coeff[s, a, :nn - 1]) and np.array_equal(res.path_param[s, a, :npt], res2.path_param[s, a, :npt])
        assert np.array_equal(vres.vx[s, a, :npt], vres2.vx[s, a, :npt]) and np.array_equal(vres.ax[s, a, :npt], vres2.ax[s, a, :npt])
    # ... and after a run that ends on every one of the buffer sets
    for reps in (1, 2, 3, 5):
        hip_backend.batch_run(reps=reps, timed=False)
        res3, vres3 = hip_backend.batch_download()
        assert np.array_equal(res.valid, res3.valid) and np.array_equal(res.n_pts, res3.n_pts)
        for s, a in zip(*np.nonzero(res.valid)):
            npt = int(res.n_pts[s, a])
            assert np.array_equal(vres.vx[s, a, :npt], vres3.vx[s, a, :npt]) and np.array_equal(res.path_param[s, a, :npt], res3.path_param[s, a, :npt]), reps


def test_tick_paths_equal_plan_paths(monteblanco, hip_backend):
    """The fused kernel's path stage is the seam-(1) kernel body: identical bits."""
    batch, vel = make_tick_inputs(monteblanco, 256, 5, 8)
    res, _ = hip_backend.tick_batch(batch, vel)
    res1 = hip_backend.plan_paths(batch)
    for name in ("nodes", "node_idx", "coeff", "path_param", "valid", "action_id", "n_pts", "reduced", "n_ties"):
        assert np.array_equal(getattr(res, name), getattr(res1, name)), name


def test_compact_trajectories_equal_the_slab_outputs(monteblanco, hip_backend):
    """ltpl_tick_batch_compact packs exactly the rows ltpl_tick_batch returns (trimmed to max_rows), for a batch on the
    one-wave pipeline and for a small batch on the fused kernel; s is the running sum of the element lengths (OTH.py:743)."""
    from scenarios import random_scenarios
    lat = monteblanco
    for n in (96, 5):
        scen, vels = random_scenarios(lat, n, seed=31 + n)
        batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
        params = _capi.VelParamSet(len_veh=lat.veh_length)
        pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
        vel = _capi.TickVelBatch(params, n, np.full(n, 25.0), np.full(n, 25.0), pos, np.concatenate(vels))
        res, vres = hip_backend.tick_batch(batch, vel)
        comp = hip_backend.new_compact_trajectories(n, max_rows=115)
        hip_backend.tick_batch_compact(batch, vel, comp)
        assert comp.struct.total_rows == int(np.minimum(res.n_pts * res.valid, 115).sum())
        for s in range(n):
            tr = comp.trajectories(s)
            names = [_capi.ACTION_NAMES[int(res.action_id[s, a])] for a in range(int(res.n_actions[s])) if res.valid[s, a]]
            assert list(tr.keys()) == names
            for a in range(int(res.n_actions[s])):
                if not res.valid[s, a]:
                    continue
                m = min(int(res.n_pts[s, a]), 115)
                t = tr[_capi.ACTION_NAMES[int(res.action_id[s, a])]][0]
                assert t.shape == (m, 7)
                assert np.array_equal(t[:, 1:5], res.path_param[s, a, :m, 0:4])
                assert np.array_equal(t[:, 5], vres.vx[s, a, :m]) and np.array_equal(t[:, 6], vres.ax[s, a, :m])
                s_ref = np.concatenate(([0.0], np.cumsum(res.path_param[s, a, :m - 1, 4])))
                assert_close_rel(t[:, 0], s_ref, what="s column")
                assert comp.vel_bound[s * 3 + a] == vres.vel_bound[s, a] and comp.reduced[s * 3 + a] == res.reduced[s, a]


def test_follow_jobs_finished_by_the_lane_kernel_or_by_the_final_kernel(monteblanco, hip_backend, monkeypatch):
    """Round 6: in large batches the lane kernel finishes a follow job itself (controlled part and unconstrained profile in one lane, capped
    backward sweep); in smaller ones the two halves run on two waves and k_vel_final composes. Same operations on the same values: every
    output of a 600-scenario batch with an opponent ahead in every scenario is IDENTICAL bit for bit between a handle that takes the first
    route (the suite's setting) and one that takes the second."""
    from graphbasedlocaltrajectoryplanner_amd.scenario_gen import c2_scenarios
    n = 600
    scen, vels = c2_scenarios(monteblanco, n, seed=77, lead_gap=(20.0, 80.0))
    rng = np.random.default_rng(5)
    params = _capi.VelParamSet(len_veh=monteblanco.veh_length)
    vplan = rng.uniform(5.0, 60.0, n)
    pos = np.array([monteblanco.node_pos[monteblanco.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    vel = _capi.TickVelBatch(params, n, vplan, vplan, pos, np.concatenate(vels))
    monkeypatch.setenv("LTPL_FOLLOW_EMIT_MIN_SCEN", "1000000")
    other = _capi.HipBackend(monteblanco)
    (ra, va), (rb, vb) = hip_backend.tick_batch(batch, vel), other.tick_batch(batch, vel)
    assert np.array_equal(ra.n_actions, rb.n_actions) and np.array_equal(va.vel_bound, vb.vel_bound) and np.array_equal(va.too_close, vb.too_close)
    n_follow = 0
    for s in range(n):
        for k in range(int(ra.n_actions[s])):
            if ra.valid[s, k]:
                m = int(ra.n_pts[s, k])
                n_follow += int(ra.action_id[s, k] == _capi.ACT_FOLLOW)
                assert np.array_equal(va.vx[s, k, :m], vb.vx[s, k, :m]) and np.array_equal(va.ax[s, k, :m], vb.ax[s, k, :m]), (s, k)
    assert n_follow >= n // 2, n_follow
    other.close()


def test_large_batches_are_planned_in_the_order_of_their_start_layers_with_identical_results(monteblanco, hip_backend, monkeypatch):
    """Round 6: batches of >= 2 048 scenarios are planned sorted by start layer (block b -> scenario order[b]); outputs are indexed by scenario,
    so every output must be IDENTICAL bit for bit to a handle that plans them in the caller's order (LTPL_NO_SCEN_ORDER=1) -- 2 500 scenarios
    with shuffled start layers, a zero-vehicle scenario and repeated start layers among them."""
    n = 2500
    batch, vel = make_tick_inputs(monteblanco, n, seed=41)
    monkeypatch.setenv("LTPL_NO_SCEN_ORDER", "1")
    other = _capi.HipBackend(monteblanco)
    (ra, va), (rb, vb) = hip_backend.tick_batch(batch, vel), other.tick_batch(batch, vel)
    for name in ("n_actions", "end_layer", "closest_obj_index", "closest_obj_node"):
        assert np.array_equal(getattr(ra, name), getattr(rb, name)), name
    assert np.array_equal(va.vel_bound, vb.vel_bound) and np.array_equal(va.too_close, vb.too_close)
    for s in range(n):
        na = int(ra.n_actions[s])
        for name in ("action_id", "valid", "reduced", "goal_layer", "n_nodes", "n_pts", "n_ties"):
            assert np.array_equal(getattr(ra, name)[s, :na], getattr(rb, name)[s, :na]), (s, name)
        for k in range(na):
            if ra.valid[s, k]:
                m, nn = int(ra.n_pts[s, k]), int(ra.n_nodes[s, k])
                assert np.array_equal(ra.nodes[s, k, :nn], rb.nodes[s, k, :nn]) and np.array_equal(ra.coeff[s, k, :nn - 1], rb.coeff[s, k, :nn - 1]), (s, k)
                assert np.array_equal(ra.path_param[s, k, :m], rb.path_param[s, k, :m]), (s, k)
                assert np.array_equal(va.vx[s, k, :m], vb.vx[s, k, :m]) and np.array_equal(va.ax[s, k, :m], vb.ax[s, k, :m]), (s, k)
    other.close()
