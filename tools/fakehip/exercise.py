"""TEST TOOL: drives every entry point of the C ABI through the HOST code of libltpl_hip.so built against the stand-in runtime
(tools/fakehip/build.sh) under AddressSanitizer / UBSan. Kernels do nothing there, so results are all-zero and are NOT looked at: the
point is that every pack / size / copy / scatter of the host side runs, on every lattice fixture and across the batch-size switches,
with the sanitizer watching both ends of every transfer. Run through tools/fakehip/run.sh."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from graphbasedlocaltrajectoryplanner_amd import _capi                      # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice            # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import random_scenarios, c2_scenarios   # noqa: E402
from test_gpu_vel import random_jobs                      # noqa: E402
from test_gpu_edge_cases import crowded                                     # noqa: E402

FAKE = os.path.join(ROOT, "tools", "fakehip", "build", "libltpl_hip_fake.so")
W = [0.0, 0.5, 0.8]
calls = [0]


def expect_error(fn, what):
    try:
        fn()
    except _capi.BackendError as e:
        print("    (refused as expected: %s: %s)" % (what, str(e)[:90]))
        return
    raise AssertionError("no error for " + what)


GARBAGE = os.environ.get("FAKEHIP_GARBAGE") is not None


def tolerant(fn):
    """FAKEHIP_GARBAGE mode: launches leave random words behind, so anything that INTERPRETS results may fail with a Python error or a
    status code -- that is fine; what must not happen is a sanitizer report from the library's host code."""
    if not GARBAGE:
        return fn()
    try:
        return fn()
    except (KeyError, ValueError, IndexError, OverflowError, MemoryError, _capi.BackendError) as e:
        return None


def exercise(name, lat):
    hip = _capi.HipBackend(lat, lib_path=FAKE)
    print("%s: L=%d V=%d E=%d, caps nodes %d pts %d" % (name, lat.num_layers, lat.num_nodes, lat.num_edges,
                                                       hip.caps.max_path_nodes, hip.caps.max_path_pts))
    rng = np.random.default_rng(5)
    # seam (1): batch sizes around every switch (fused <= 8, four-wave, one-wave batch kernel, pipeline)
    for n, n_veh in ((1, 8), (2, 0), (7, 3), (8, 8), (9, 16), (63, 8), (64, 8), (65, 1), (300, 8), (1500, 4)):
        scen, vels = random_scenarios(lat, n, seed=n, n_veh=n_veh)
        if not lat.closed:                                     # the last layers of an open track have no planning range
            for sc in scen:
                sl = min(sc['start_node'][0], lat.num_layers - 8)
                sc['start_node'] = (sl, int(lat.raceline_index[sl]))
                sc['last_nodes'] = None
        batch = _capi.PathsBatch(scen, w_last_edges=W)
        res = hip.plan_paths(batch)
        tolerant(lambda: res.action_sets(0, scen[0]['start_node'][0], lat.num_layers))
        tolerant(lambda: hip.plan_paths_mask(batch))           # + the blocked-edge export (sweep order -> CSC order on the host)
        vplan = rng.uniform(0.0, 60.0, n)
        pos = np.array([lat.node_pos[lat.layer_off[sc['start_node'][0]] + sc['start_node'][1]] for sc in scen])
        vt = _capi.TickVelBatch(_capi.VelParamSet(len_veh=lat.veh_length), n, vplan, vplan + 0.5, pos,
                                np.concatenate(vels) if n_veh else np.zeros(0))
        bt = batch
        r, v = hip.tick_batch(bt, vt)
        comp = hip.new_compact_trajectories(n, max_rows=115 if n % 2 else 0)
        tolerant(lambda: (hip.tick_batch_compact(bt, vt, comp), comp.trajectories(n - 1)))
        hip.batch_upload(bt, vt)
        hip.batch_run(reps=2, timed=True)
        hip.batch_last_paths_ms()
        hip.batch_run_profile(reps=2)
        hip.batch_download()
        calls[0] += 9
    if name == "monteblanco":
        # capacity maxima of one scenario: 96 vehicles / 192 positions; and one over
        hip.plan_paths(_capi.PathsBatch(crowded(lat, 3, 96, 1, seed=1), w_last_edges=W))
        hip.plan_paths(_capi.PathsBatch(crowded(lat, 3, 12, 15, seed=2), w_last_edges=W))
        expect_error(lambda: hip.plan_paths(_capi.PathsBatch(crowded(lat, 1, 100, 0, seed=3), w_last_edges=W)), "100 vehicles")
        scen, _ = c2_scenarios(lat, 2, seed=3)
        bad = dict(scen[0]); bad["start_node"] = (lat.num_layers + 5, 0)
        expect_error(lambda: hip.plan_paths(_capi.PathsBatch([bad], w_last_edges=W)), "start layer out of range")
        params = _capi.VelParamSet(len_veh=lat.veh_length)
        expect_error(lambda: hip.vel_profile(params, [{"mode": _capi.VEL_FB, "kappa": np.zeros(5), "el_lengths": np.ones(5),
                                                       "loc_gg": np.ones((5, 2)) * 5.0, "v_start": 10.0, "v_end": 5.0}]), "el_lengths")
    # seam (2): job counts around the zero-copy switch (<= 16) and large, every variant of the kernel
    for exp, axm, varying in ((1.0, None, False), (2.0, np.array([[0.0, 6.0], [30.0, 4.0], [80.0, 1.0]]), True), (1.5, None, True)):
        params = _capi.VelParamSet(dyn_model_exp=exp, len_veh=lat.veh_length) if axm is None else \
            _capi.VelParamSet(dyn_model_exp=exp, len_veh=lat.veh_length, ax_max_machines=axm)
        for nj in (1, 16, 17, 250):
            tolerant(lambda: hip.vel_profile(params, random_jobs(lat, rng, nj, varying)))
            calls[0] += 1
    # object ingestion, race-line projection, constant-segment test
    n = 5000
    l = rng.integers(0, lat.num_layers, n)
    p = lat.refline[l] + lat.normvec[l] * rng.uniform(-8.0, 8.0, n)[:, None]
    tolerant(lambda: hip.process_objects(p[:, 0], p[:, 1], rng.uniform(-3, 3, n), rng.uniform(0, 80, n), rng.uniform(3, 6, n)))
    hip.process_objects(np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0))
    tolerant(lambda: hip.raceline_s(p[3]))
    seg = np.column_stack((lat.refline[:30], np.zeros(30), np.zeros(30), np.ones(30)))
    tolerant(lambda: hip.const_segment_test(seg, p[0], [(2.5, p[:3])]))
    calls[0] += 4
    # the planner entry points: start pose, one tick (no path comes back from a kernel that does nothing -> the state machine must say so)
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    for n_scen in (1, 3, 70):
        pl = Planner(hip, n_scen)
        sl = 5
        pos = lat.node_pos[lat.layer_off[sl] + lat.raceline_index[sl]]
        for s in range(n_scen):
            pl.set_start(s, pos, float(lat.node_psi[lat.layer_off[sl] + lat.raceline_index[sl]]), 0.0)
        veh = [[(2.5, 10.0, p[:2])] for _ in range(n_scen)]
        try:
            pl.calc_paths(["straight"] * n_scen, 0.0, veh, None)
            pl.paths(0)
            pl.calc_vel_profile([pos] * n_scen, 0.0)
            pl.trajectories(n_scen - 1)
        except (_capi.BackendError, KeyError, ValueError, IndexError) as e:
            print("    (planner with empty / garbage kernel results: %s)" % str(e)[:100])
        pl.close()
        calls[0] += 4
    # the fleet entry points (state in "device" memory; the kernels do nothing here, so every planner keeps its start state): packing of
    # the per-call inputs, the grouped tape inputs, the query views
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    for n_pl in (1, 70):
        fl = Fleet(hip, n_pl)
        sl = 5
        g0 = lat.layer_off[sl] + lat.raceline_index[sl]
        pos = lat.node_pos[g0]
        for s in range(n_pl):
            fl.set_start(s, pos, float(lat.node_psi[g0]), 0.0)
        fl.set_start_range(0, n_pl, pos, float(lat.node_psi[g0]), 0.0)          # (the one-call form over the whole fleet)
        if n_pl > 3:
            fl.set_start_range(2, n_pl - 1, pos, float(lat.node_psi[g0]), 1.0)
        veh = [[(2.5, 10.0, p[:2]), (2.0, 3.0, p[2:3])] for _ in range(n_pl)]
        try:
            fl.calc_paths(["straight"] * n_pl, 0.0, veh, [[1, 2, 3]] * n_pl)
            fl.paths(0)
            fl.calc_vel_profile([pos] * n_pl, 0.0, incl_emerg_traj=True)
            fl.trajectories(n_pl - 1)
            fl.calc_paths_begin(["straight"] * n_pl, 0.0, veh)
            fl.calc_paths_finish([[4, 5]] * n_pl)
            fl.get_ref_idx([pos] * n_pl, scen=0)
            grp = dict(prev_action="straight", t_now=0.1, vehicles=veh[0], zone_gids=[7, 8], pos_est=pos, vel_est=1.0, incl_emerg_traj=True)
            for k in range(3):
                fl.tape_append_groups([(n_pl - n_pl // 2, grp), (n_pl // 2, dict(grp, vehicles=[]))] if n_pl > 1 else [(1, grp)])
            fl.tape_append(["straight"] * n_pl, 0.2, veh, None, [pos] * n_pl, 0.0)
            fl.tape_run(0, 4)
            fl.tape_clear()
        except (_capi.BackendError, KeyError, ValueError, IndexError) as e:
            print("    (fleet with empty kernel results: %s)" % str(e)[:100])
        fl.close()
        calls[0] += 12
    hip.close()


def create_failures(lat):
    """ltpl_create with the n-th device allocation failing: every error return releases what was acquired so far (no leak / double free
    under the sanitizers), the error is reported, and a later create works."""
    import ctypes as C
    lib = C.CDLL(FAKE)
    lib.fakehip_fail_malloc_after.argtypes = [C.c_long]
    refused = 0
    for n in (1, 2, 3, 7, 15, 25, 33, 40, 45, 60):
        lib.fakehip_fail_malloc_after(n)
        try:
            _capi.HipBackend(lat, lib_path=FAKE).close()
        except _capi.BackendError:
            refused += 1
        lib.fakehip_fail_malloc_after(0)
    _capi.HipBackend(lat, lib_path=FAKE).close()
    print("create with injected allocation failures: %d of 10 refused, create works afterwards" % refused)
    assert refused >= 5
    # growth of the staging buffers fails in the middle of a call: error now, and the handle works on the next call
    hip = _capi.HipBackend(lat, lib_path=FAKE)
    grown = 0
    for n in (10, 200, 3000):
        scen, vels = random_scenarios(lat, n, seed=n)
        batch = _capi.PathsBatch(scen, w_last_edges=W)
        vt = _capi.TickVelBatch(_capi.VelParamSet(len_veh=lat.veh_length), n, np.full(n, 20.0), np.full(n, 20.0),
                                np.array([lat.node_pos[lat.layer_off[sc['start_node'][0]] + sc['start_node'][1]] for sc in scen]), np.concatenate(vels))
        for entry in (lambda: hip.plan_paths(batch), lambda: hip.tick_batch(batch, vt), lambda: hip.batch_upload(batch, vt)):
            lib.fakehip_fail_malloc_after(1)
            try:
                entry()
            except _capi.BackendError:
                grown += 1
            lib.fakehip_fail_malloc_after(0)
            entry()
    hip.close()
    print("staging growth with injected allocation failures: %d calls refused, every retry worked" % grown)
    assert grown >= 3
    calls[0] += 30


def main():
    g = os.path.join(ROOT, "tests", "golden")
    create_failures(Lattice.load(os.path.join(g, "millbrook_lattice.npz")))
    quick = "--quick" in sys.argv                              # (the CPU test suite's leg: two lattices)
    for name in ("monteblanco", "millbrook") if quick else ("monteblanco", "open", "zalazone", "millbrook", "lvms"):
        exercise(name, Lattice.load(os.path.join(g, name + "_lattice.npz")))
    if not quick:
        from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice
        exercise("c3 synthetic", c3_lattice())
    # offline edge evaluation entry point
    from graphbasedlocaltrajectoryplanner_amd import offline_build as ob
    import ctypes as C
    with np.load(os.path.join(g, "zalazone_track.npz")) as z:
        track = {k: z[k] for k in z.files}
    try:
        ob.build_lattice(track, ob.OFFLINE_DEFAULTS, ob.edges_on_device(C.CDLL(FAKE)))
    except Exception as e:                                     # all-zero edge results cannot give a lattice; the call itself ran
        print("offline build on empty kernel results: %s: %s" % (type(e).__name__, str(e)[:100]))
    print("done: %d entry-point calls" % calls[0])


if __name__ == "__main__":
    main()
