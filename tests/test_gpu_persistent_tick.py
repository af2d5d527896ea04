"""
GPU: the PERSISTENT single tick (k_tick_persistent, include/ltpl_hip.h ABI v9: ltpl_create_ex(LTPL_CREATE_PERSISTENT_TICK)) -- a resident
workgroup that receives every single-scenario ltpl_tick_batch call through a mailbox in page-locked memory instead of being launched.
Same device code as the launched k_tick, so every output must be IDENTICAL bit for bit; what is new is the life cycle of the kernel
(start at the first tick, leave when another entry point needs the handle / on request / after the idle limit, start again), and
that is what most of these tests exercise. Every test bounds the residency (idle limit of a few ms .. 250 ms): a test that fails cannot
leave a kernel spinning on the box.
"""
import time

import numpy as np
import pytest

from graphbasedlocaltrajectoryplanner_amd import _capi

pytestmark = pytest.mark.gpu

W_LAST = [0.0, 0.5, 0.8]


def single_ticks(lat, n, seed):
    from scenarios import random_scenarios
    scen, vels = random_scenarios(lat, n, seed=seed)
    params = _capi.VelParamSet(len_veh=lat.veh_length)
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        batch = _capi.PathsBatch(scen[i:i + 1], w_last_edges=W_LAST)
        sl, sn = scen[i]['start_node']
        pos = lat.node_pos[lat.layer_off[sl] + sn][None, :]
        vp = float(rng.uniform(0.0, 55.0))
        out.append((batch, _capi.TickVelBatch(params, 1, np.full(1, vp), np.full(1, vp + 0.3), pos, vels[i])))
    return out


def assert_same_tick(a, b, what):
    (ra, va), (rb, vb) = a, b
    for name in ("n_actions", "end_layer", "closest_obj_index"):
        assert np.array_equal(getattr(ra, name), getattr(rb, name)), (what, name)
    na = int(ra.n_actions[0])
    for name in ("action_id", "valid", "reduced", "goal_layer", "n_nodes", "n_pts", "n_ties"):
        assert np.array_equal(getattr(ra, name)[0, :na], getattr(rb, name)[0, :na]), (what, name)
    assert np.array_equal(va.vel_bound[0, :na], vb.vel_bound[0, :na]) and np.array_equal(va.too_close[0, :na], vb.too_close[0, :na]), what
    for k in range(na):
        if ra.valid[0, k]:
            n, nn = int(ra.n_pts[0, k]), int(ra.n_nodes[0, k])
            assert np.array_equal(ra.nodes[0, k, :nn], rb.nodes[0, k, :nn]) and np.array_equal(ra.node_idx[0, k, :nn], rb.node_idx[0, k, :nn]), (what, k)
            assert np.array_equal(ra.coeff[0, k, :nn - 1], rb.coeff[0, k, :nn - 1]), (what, k, "coeff")
            assert np.array_equal(ra.path_param[0, k, :n], rb.path_param[0, k, :n]), (what, k, "path_param")
            assert np.array_equal(va.vx[0, k, :n], vb.vx[0, k, :n]) and np.array_equal(va.ax[0, k, :n], vb.ax[0, k, :n]), (what, k, "vx / ax")


def assert_same_paths(ra, rb, what):
    """Seam-(1) outputs of two handles: every scenario, every offered slot (entries behind n_actions / n_nodes / n_pts are unspecified
    padding of the caller's slabs)."""
    assert np.array_equal(ra.n_actions, rb.n_actions), what
    for s in range(ra.n_scen):
        na = int(ra.n_actions[s])
        for name in ("action_id", "valid", "n_nodes", "n_pts"):
            assert np.array_equal(getattr(ra, name)[s, :na], getattr(rb, name)[s, :na]), (what, s, name)
        for k in range(na):
            if ra.valid[s, k]:
                n, nn = int(ra.n_pts[s, k]), int(ra.n_nodes[s, k])
                assert np.array_equal(ra.nodes[s, k, :nn], rb.nodes[s, k, :nn]) and np.array_equal(ra.path_param[s, k, :n], rb.path_param[s, k, :n]), (what, s, k)


def device_synchronize():
    """hipDeviceSynchronize of the HIP runtime the library itself uses (the process holds ONE runtime: torch's bundled copy, initialised
    second, would not find the device)."""
    import ctypes
    with open("/proc/self/maps") as fh:                       # the copy libltpl_hip.so pulled in, by its path
        paths = sorted({l.split()[-1] for l in fh if "libamdhip64" in l})
    assert paths, "libltpl_hip.so is loaded, so is its HIP runtime"
    rt = ctypes.CDLL(paths[0])
    assert rt.hipDeviceSynchronize() == 0


class _Arrays(object):
    """The output arrays of a result object, copied (the objects themselves hold ctypes pointers and cannot be copied)."""

    def __init__(self, obj, names):
        for name in names:
            setattr(self, name, np.array(getattr(obj, name), copy=True))


def copy_of(res_vres):
    res, vres = res_vres
    return (_Arrays(res, ("n_actions", "end_layer", "closest_obj_index", "action_id", "valid", "reduced", "goal_layer", "n_nodes", "n_pts", "n_ties",
                          "nodes", "node_idx", "coeff", "path_param")),
            _Arrays(vres, ("vx", "ax", "vel_bound", "too_close")))


def test_resident_kernel_gives_the_launched_kernels_results_bit_for_bit(monteblanco, hip_backend, monkeypatch):
    monkeypatch.setenv("LTPL_PERSIST_IDLE_MS", "100")
    pers = _capi.HipBackend(monteblanco, persistent_tick=True)
    st = pers.persistent_stats()
    assert st["enabled"] == 1 and st["resident"] == 0 and st["ticks"] == 0 and st["idle_ms"] == 100.0
    ticks = single_ticks(monteblanco, 48, seed=5)
    n_follow = 0
    for i, (b, v) in enumerate(ticks):
        got = pers.tick_batch(b, v)
        exp = hip_backend.tick_batch(b, v)
        assert_same_tick(got, exp, "tick %d" % i)
        n_follow += int(((got[0].action_id == _capi.ACT_FOLLOW) & (got[0].valid == 1)).sum())
    st = pers.persistent_stats()
    assert st["ticks"] == 48 and st["launches"] == 1 and st["resident"] == 1, st
    assert 0.0 < st["device_us_mean"] < 5000.0 and st["device_us_last"] > 0.0, st
    assert n_follow >= 3
    pers.close()                                               # (a resident kernel: ltpl_destroy makes it leave first)
    device_synchronize()                                       # nothing of it is left on the device


def test_other_entry_points_of_the_handle_stop_and_restart_the_resident_kernel(monteblanco, hip_backend, monkeypatch):
    """Seam (1) on its own, a batch of scenarios (the pipeline), the resident batch of the benchmark and a two-scenario tick all reuse the
    handle's staging buffers: the resident kernel leaves before they run and the next single tick starts it again -- results identical
    throughout, the launch counter tells the story."""
    from scenarios import random_scenarios
    monkeypatch.setenv("LTPL_PERSIST_IDLE_MS", "100")
    pers = _capi.HipBackend(monteblanco, persistent_tick=True)
    ticks = single_ticks(monteblanco, 12, seed=9)
    scen, vels = random_scenarios(monteblanco, 96, seed=10)
    params = _capi.VelParamSet(len_veh=monteblanco.veh_length)
    pos = np.array([monteblanco.node_pos[monteblanco.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    big = (_capi.PathsBatch(scen, w_last_edges=W_LAST), _capi.TickVelBatch(params, 96, np.full(96, 25.0), np.full(96, 25.0), pos, np.concatenate(vels)))
    two = (_capi.PathsBatch(scen[:2], w_last_edges=W_LAST), _capi.TickVelBatch(params, 2, np.full(2, 25.0), np.full(2, 25.0), pos[:2], np.concatenate(vels[:2])))
    launches = 0
    for i, (b, v) in enumerate(ticks):
        assert_same_tick(pers.tick_batch(b, v), hip_backend.tick_batch(b, v), "tick %d" % i)
        launches += 1 if i % 3 == 0 else 0
        assert pers.persistent_stats()["launches"] == launches, (i, pers.persistent_stats())
        if i % 3 == 2:                                          # another entry point every third tick
            k = i // 3
            if k == 0:
                a, e = pers.plan_paths(b), hip_backend.plan_paths(b)
                assert_same_paths(a, e, "seam (1) between ticks")
            elif k == 1:
                (ra, va), (re_, ve) = pers.tick_batch(*big), hip_backend.tick_batch(*big)
                assert_same_paths(ra, re_, "batch between ticks")
                assert np.array_equal(va.vel_bound, ve.vel_bound)
            elif k == 2:
                pers.batch_upload(*big); pers.batch_run(reps=2, timed=False); pers.batch_download()
            else:
                (ra, va), (re_, ve) = pers.tick_batch(*two), hip_backend.tick_batch(*two)
                assert_same_paths(ra, re_, "two scenarios between ticks")
                assert np.array_equal(va.vel_bound, ve.vel_bound)
            assert pers.persistent_stats()["resident"] == 0
    assert pers.persistent_stats()["ticks"] == 12
    pers.close()


def test_the_resident_kernel_leaves_by_itself_and_comes_back(monteblanco, hip_backend, monkeypatch):
    """Idle limit 5 ms: after a pause the kernel has left (nothing resident, a device-wide synchronisation returns at once); the next
    tick starts it again; ticks posted right at the limit -- the race between the kernel's last poll and the host's post -- are served."""
    monkeypatch.setenv("LTPL_PERSIST_IDLE_MS", "5")
    pers = _capi.HipBackend(monteblanco, persistent_tick=True)
    ticks = single_ticks(monteblanco, 40, seed=21)
    exp = [copy_of(hip_backend.tick_batch(b, v)) for b, v in ticks]
    assert_same_tick(pers.tick_batch(*ticks[0]), exp[0], "first")
    time.sleep(0.05)
    assert pers.persistent_stats()["resident"] == 0
    t0 = time.perf_counter(); device_synchronize(); assert time.perf_counter() - t0 < 0.05
    assert_same_tick(pers.tick_batch(*ticks[1]), exp[1], "after the pause")
    assert pers.persistent_stats()["launches"] == 2
    rng = np.random.default_rng(3)
    for i in range(2, 40):                                      # pauses around the limit
        time.sleep(float(rng.uniform(0.0035, 0.0065)))
        assert_same_tick(pers.tick_batch(*ticks[i]), exp[i], "tick %d" % i)
    st = pers.persistent_stats()
    assert st["ticks"] == 40 and 2 <= st["launches"] <= 40, st
    pers.close()


def test_stop_on_request_and_device_wide_synchronisation(monteblanco, hip_backend, monkeypatch):
    monkeypatch.setenv("LTPL_PERSIST_IDLE_MS", "250")
    pers = _capi.HipBackend(monteblanco, persistent_tick=True)
    b, v = single_ticks(monteblanco, 1, seed=2)[0]
    assert_same_tick(pers.tick_batch(b, v), hip_backend.tick_batch(b, v), "tick")
    assert pers.persistent_stats()["resident"] == 1
    pers.persistent_stop()
    assert pers.persistent_stats()["resident"] == 0
    t0 = time.perf_counter(); device_synchronize(); assert time.perf_counter() - t0 < 0.1
    pers.persistent_stop()                                      # no-op
    assert_same_tick(pers.tick_batch(b, v), hip_backend.tick_batch(b, v), "tick after the stop")
    # a device-wide synchronisation WITH a resident kernel waits for the idle limit, not for ever
    t0 = time.perf_counter(); device_synchronize(); waited = time.perf_counter() - t0
    assert waited < 2.0 and pers.persistent_stats()["resident"] == 0, waited
    pers.close()


def test_other_kernel_variants_and_lattices(monteblanco, hip_backend, monkeypatch):
    """A tick with another machine table / exponent is another kernel variant: the resident kernel is replaced. A lattice without a
    compile-time plan for the four-wave kernel does not engage the mode (stats say so) and behaves like ltpl_create."""
    from test_other_tracks import lattice_of
    monkeypatch.setenv("LTPL_PERSIST_IDLE_MS", "100")
    pers = _capi.HipBackend(monteblanco, persistent_tick=True)
    (b, v) = single_ticks(monteblanco, 1, seed=4)[0]
    table = _capi.VelParamSet(len_veh=monteblanco.veh_length, ax_max_machines=[[0.0, 6.0], [36.0, 6.0], [72.0, 2.5]], dyn_model_exp=2.0)
    v2 = _capi.TickVelBatch(table, 1, v.vel_plan, v.vel_est, np.column_stack((v.pos_x, v.pos_y)), v.veh_vel)
    for k, vv in enumerate((v, v2, v, v2)):
        assert_same_tick(pers.tick_batch(b, vv), hip_backend.tick_batch(b, vv), "variant %d" % k)
    assert pers.persistent_stats()["launches"] == 4
    pers.close()
    lat = lattice_of("berlin")                                  # plan class C: runtime plan for the four-wave kernel
    other = _capi.HipBackend(lat, persistent_tick=True)
    plain = _capi.HipBackend(lat)
    assert other.persistent_stats()["enabled"] == 0
    (b, v) = single_ticks(lat, 1, seed=4)[0]
    assert_same_tick(other.tick_batch(b, v), plain.tick_batch(b, v), "berlin")
    assert other.persistent_stats()["launches"] == 0
    other.close(); plain.close()
