"""GPU: the BASELINE configurations that are parity-test cases rather than bench lines.

  C3  synthetic 10 000-node / 98 000-edge oval, 32 static obstacles (64 obstacle positions) per scenario
  C4  1 024 independent C2 scenarios, block-sharded into 8 parts (the per-GPU shards of an 8 x MI355X node are run one
      after the other on the single GPU of the test box): identical bits to the unsharded batch
  C5  high-resolution oval (0.5 m layer spacing, 21 lateral nodes) with a slow opponent ahead: the single-tick latency
      kernel incl. the follow-mode velocity profile on every tick
All through the C ABI (libltpl_hip.so) against the oracle on identical packed inputs."""
import os

import numpy as np
import pytest

from test_gpu_paths import compare_results
from test_gpu_vel import compare_tick
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import c2_scenarios, raceline_state
from graphbasedlocaltrajectoryplanner_amd.sharding import shard_bounds, RESULT_FIELDS, VEL_FIELDS
from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice, c5_lattice, scattered_obstacle_scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    from oracle.oracle_lib import OracleBackend
    lat = c3_lattice()
    return lat, _capi.HipBackend(lat), OracleBackend(lat)


def vel_inputs(lat, scen, vels, seed):
    n = len(scen)
    rng = np.random.default_rng(seed)
    vplan = rng.uniform(3.0, 50.0, n)
    pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    params = _capi.VelParamSet(len_veh=lat.veh_length)
    return _capi.TickVelBatch(params, n, vplan, vplan, pos, np.concatenate(vels) if len(vels) else np.zeros(0))


def test_c3_synthetic_lattice_matches_oracle(c3):
    lat, hip, orc = c3
    assert lat.num_nodes == 10000 and 90000 < lat.num_edges < 110000
    scen, vels = scattered_obstacle_scenarios(lat, 256, n_obj=32, seed=0)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    assert int(batch.pos_off[-1]) == 256 * 64                      # O = 64 obstacle positions per scenario
    res, ref = hip.plan_paths(batch), orc.plan_paths(batch)
    compare_results(res, ref, lat)
    assert int(res.n_ties.sum()) > 0                               # the symmetric oval produces exact cost ties
    names = set(int(x) for x in res.action_id[res.valid == 1])
    assert {_capi.ACT_FOLLOW, _capi.ACT_LEFT, _capi.ACT_RIGHT} <= names
    vel = vel_inputs(lat, scen, vels, 3)
    r2, v2 = hip.tick_batch(batch, vel)
    o2, ov2 = orc.tick_batch(batch, vel)
    compare_tick(r2, v2, o2, ov2)
    # single-scenario (latency kernel) launches give the same bits as the batch kernel
    for i in (0, 17, 101):
        b1 = _capi.PathsBatch([scen[i]], w_last_edges=[0.0, 0.5, 0.8])
        r1 = hip.plan_paths(b1)
        for name in ("valid", "action_id", "n_pts", "n_nodes", "reduced", "n_ties"):
            assert np.array_equal(getattr(r1, name)[0], getattr(res, name)[i]), name
        for a in range(3):
            if res.valid[i, a]:
                nn, npts = int(res.n_nodes[i, a]), int(res.n_pts[i, a])
                assert np.array_equal(r1.nodes[0, a, :nn], res.nodes[i, a, :nn])
                assert np.array_equal(r1.path_param[0, a, :npts], res.path_param[i, a, :npts])


@pytest.mark.parametrize("env", [{"LTPL_NO_FIXED_PLAN": "1"},
                                 {"LTPL_LDS_POISON": "0xfff80000"},
                                 {"LTPL_LDS_POISON": "0x00000001", "LTPL_NO_FIXED_PLAN": "1"},
                                 {"LTPL_FORCE_LONG_HORIZON": "1"}])
def test_batch_kernel_plan_classes_and_stale_lds(env, monteblanco, oracle_backend, monkeypatch):
    """The one-wave batch kernel exists in compile-time LDS plan classes and with a runtime plan (any lattice): both must
    give the same bits; so must the long-horizon mode (parent tables in global memory) when it is forced on a lattice that
    would fit. LTPL_LDS_POISON fills the team's LDS with a word before phase 0 -- a scenario must not depend on
    what an earlier workgroup left behind."""
    from oracle.oracle_lib import OracleBackend
    from scenarios import random_scenarios
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lat3 = c3_lattice()
    for lat, orc, scen in ((monteblanco, oracle_backend, random_scenarios(monteblanco, 192, seed=7, n_veh=8)[0]),
                           (lat3, OracleBackend(lat3), scattered_obstacle_scenarios(lat3, 128, n_obj=32, seed=3)[0])):
        # environment is read at ltpl_create; LTPL_LDS_POISON only exists in the experiment build of the library
        hip = _capi.HipBackend(lat, lib_path=_capi.experiment_library_path() if "LTPL_LDS_POISON" in env else None)
        batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
        compare_results(hip.plan_paths(batch), orc.plan_paths(batch), lat)


def test_c4_sharded_batch_is_bit_identical(monteblanco, hip_backend):
    lat = monteblanco
    n, world = 1024, 8
    scen, vels = c2_scenarios(lat, n, seed=1)
    vel_all = vel_inputs(lat, scen, vels, 5)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    full, vfull = hip_backend.tick_batch(batch, vel_all)
    for rank in range(world):
        lo, hi = shard_bounds(n, rank, world)
        assert hi - lo == 128
        b = _capi.PathsBatch(scen[lo:hi], w_last_edges=[0.0, 0.5, 0.8])
        v = _capi.TickVelBatch(vel_all.params, hi - lo, vel_all.vel_plan[lo:hi], vel_all.vel_est[lo:hi],
                               np.column_stack((vel_all.pos_x[lo:hi], vel_all.pos_y[lo:hi])), np.concatenate(vels[lo:hi]))
        part, vpart = hip_backend.tick_batch(b, v)
        # per-scenario / per-slot scalars: identical; slabs: identical on the rows the call defines (capacity padding and
        # slots without a path are not written by the library)
        for k in ("end_layer", "closest_obj_index", "closest_obj_node", "n_actions", "action_id", "valid", "reduced",
                  "goal_layer", "n_nodes", "n_pts", "n_ties"):
            assert np.array_equal(getattr(part, k), getattr(full, k)[lo:hi]), k
        assert np.array_equal(vpart.vel_bound, vfull.vel_bound[lo:hi]) and np.array_equal(vpart.too_close, vfull.too_close[lo:hi])
        for s_ in range(hi - lo):
            for a in range(3):
                if not part.valid[s_, a]:
                    continue
                nn, npts = int(part.n_nodes[s_, a]), int(part.n_pts[s_, a])
                assert np.array_equal(part.nodes[s_, a, :nn], full.nodes[lo + s_, a, :nn])
                assert np.array_equal(part.node_idx[s_, a, :nn], full.node_idx[lo + s_, a, :nn])
                assert np.array_equal(part.coeff[s_, a, :nn - 1], full.coeff[lo + s_, a, :nn - 1])
                assert np.array_equal(part.path_param[s_, a, :npts], full.path_param[lo + s_, a, :npts])
                assert np.array_equal(vpart.vx[s_, a, :npts], vfull.vx[lo + s_, a, :npts])
                assert np.array_equal(vpart.ax[s_, a, :npts], vfull.ax[lo + s_, a, :npts])
    assert int(full.valid.sum()) >= n


@pytest.mark.parametrize("n", [128, 513])
def test_small_batches_fused_by_default_match_the_pipeline(monteblanco, hip_backend, monkeypatch, n):
    """BASELINE config C4's shard is 128 scenarios per GPU. A handle created WITHOUT the suite's LTPL_PIPELINE_MIN_SCEN=64 serves batches up to
    two workgroups per compute unit with the fused tick kernel (one launch; 81 against 110 us per step at 128 scenarios, profiles/
    r06h_c4_fused_ab.txt) and larger ones with the pipeline: either way the outputs are those of the suite's pipeline handle -- integers
    identical, floats to 1e-9 (the two velocity formulations order their fp64 arithmetic differently)."""
    from test_gpu_vel import make_tick_inputs
    monkeypatch.delenv("LTPL_PIPELINE_MIN_SCEN", raising=False)
    dflt = _capi.HipBackend(monteblanco)
    batch, vel = make_tick_inputs(monteblanco, n, seed=31)
    (ra, va), (rb, vb) = dflt.tick_batch(batch, vel), hip_backend.tick_batch(batch, vel)
    assert np.array_equal(ra.n_actions, rb.n_actions) and np.array_equal(va.vel_bound, vb.vel_bound) and np.array_equal(va.too_close, vb.too_close)
    n_paths = 0
    for s in range(n):
        na = int(ra.n_actions[s])
        for name in ("action_id", "valid", "reduced", "n_nodes", "n_pts", "n_ties"):
            assert np.array_equal(getattr(ra, name)[s, :na], getattr(rb, name)[s, :na]), (s, name)
        for k in range(na):
            if ra.valid[s, k]:
                m, nn = int(ra.n_pts[s, k]), int(ra.n_nodes[s, k])
                n_paths += 1
                assert np.array_equal(ra.nodes[s, k, :nn], rb.nodes[s, k, :nn]) and np.array_equal(ra.path_param[s, k, :m], rb.path_param[s, k, :m]), (s, k)
                vmax = max(1.0, float(np.max(np.abs(vb.vx[s, k, :m]))))
                assert float(np.max(np.abs(va.vx[s, k, :m] - vb.vx[s, k, :m]))) <= 1e-9 * vmax, (s, k, "vx")
                assert float(np.max(np.abs(va.ax[s, k, :m] - vb.ax[s, k, :m]))) <= 1e-7 * max(5.0, vmax * vmax / 2.0), (s, k, "ax")
    assert n_paths >= n
    dflt.close()


def test_c5_highres_follow_ticks_match_oracle():
    from oracle.oracle_lib import OracleBackend
    lat = c5_lattice()
    assert lat.num_layers == 1600 and int(lat.nodes_in_layer.max()) == 21
    hip, orc = _capi.HipBackend(lat), OracleBackend(lat)
    rng = np.random.default_rng(2)
    scen, vels = [], []
    for _ in range(48):
        sl = int(rng.integers(0, lat.num_layers))
        sn = int(lat.raceline_index[sl])
        s_ego = float(lat.s_raceline[sl])
        x, y, psi, v = raceline_state(lat, s_ego + rng.uniform(20.0, 80.0))          # slow opponent ahead on the race line
        v = float(v) * rng.uniform(0.2, 0.5)
        pred = np.array([[x - np.sin(psi) * v * 0.2, y + np.cos(psi) * v * 0.2]])
        scen.append({"start_node": (sl, sn), "action_sets": True, "vehicles": [(2.5, np.vstack((np.array([[x, y]]), pred)))],
                     "zone_gids": [], "last_nodes": None, "obj_in_const": False, "obj_besides": False, "last_action": None,
                     "const_closest": None, "psi_s": float(lat.node_psi[lat.layer_off[sl] + sn])})
        vels.append(np.array([v]))
    vel = vel_inputs(lat, scen, vels, 9)
    # one call per tick = the latency kernel (one workgroup per scenario, path + velocity fused)
    n_follow = 0
    for i in range(len(scen)):
        b1 = _capi.PathsBatch([scen[i]], w_last_edges=[0.0, 0.5, 0.8])
        v1 = _capi.TickVelBatch(vel.params, 1, vel.vel_plan[i:i + 1], vel.vel_est[i:i + 1],
                                np.array([[vel.pos_x[i], vel.pos_y[i]]]), vels[i])
        r, vr = hip.tick_batch(b1, v1)
        o, ov = orc.tick_batch(b1, v1)
        compare_tick(r, vr, o, ov)
        n_follow += int(((r.action_id == _capi.ACT_FOLLOW) & (r.valid == 1)).sum())
    assert n_follow >= 24                                        # the follow-mode profile really ran
    # and the same scenarios as one batch through the throughput pipeline
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    r, vr = hip.tick_batch(batch, vel)
    o, ov = orc.tick_batch(batch, vel)
    compare_tick(r, vr, o, ov)


def c5_follow_scenarios(lat, n, seed):
    rng = np.random.default_rng(seed)
    scen, vels = [], []
    for _ in range(n):
        sl = int(rng.integers(0, lat.num_layers))
        sn = int(lat.raceline_index[sl])
        x, y, psi, v = raceline_state(lat, float(lat.s_raceline[sl]) + rng.uniform(20.0, 80.0))
        v = float(v) * rng.uniform(0.2, 0.5)
        pred = np.array([[x - np.sin(psi) * v * 0.2, y + np.cos(psi) * v * 0.2]])
        scen.append({"start_node": (sl, sn), "action_sets": True, "vehicles": [(2.5, np.vstack((np.array([[x, y]]), pred)))],
                     "zone_gids": [], "last_nodes": None, "obj_in_const": False, "obj_besides": False, "last_action": None,
                     "const_closest": None, "psi_s": float(lat.node_psi[lat.layer_off[sl] + sn])})
        vels.append(np.array([v]))
    return scen, vels


def test_c5_full_300m_horizon_long_horizon_mode():
    """C5 as BASELINE specifies it: 0.5 m layer spacing AND a 300 m horizon = 600 layers x 21 nodes. The parent tables no
    longer fit in LDS next to the path scratch: ltpl_create switches to the long-horizon mode (tables in global memory,
    velocity stage through the lane kernels). Single ticks (four-wave team) and a batch (one-wave teams) vs the oracle."""
    from oracle.oracle_lib import OracleBackend
    lat = c5_lattice(horizon=300.0)
    hip, orc = _capi.HipBackend(lat), OracleBackend(lat)
    assert hip.caps.max_path_nodes >= 601
    scen, vels = c5_follow_scenarios(lat, 80, seed=5)
    vel = vel_inputs(lat, scen, vels, 11)
    n_follow = 0
    for i in range(6):
        b1 = _capi.PathsBatch([scen[i]], w_last_edges=[0.0, 0.5, 0.8])
        v1 = _capi.TickVelBatch(vel.params, 1, vel.vel_plan[i:i + 1], vel.vel_est[i:i + 1],
                                np.array([[vel.pos_x[i], vel.pos_y[i]]]), vels[i])
        r, vr = hip.tick_batch(b1, v1)
        o, ov = orc.tick_batch(b1, v1)
        compare_tick(r, vr, o, ov)
        assert int(r.n_nodes[0].max()) >= 601
        n_follow += int(((r.action_id == _capi.ACT_FOLLOW) & (r.valid == 1)).sum())
        compare_results(hip.plan_paths(b1), orc.plan_paths(b1), lat)
    assert n_follow >= 3
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    r, vr = hip.tick_batch(batch, vel)
    o, ov = orc.tick_batch(batch, vel)
    compare_tick(r, vr, o, ov)


def test_wide_lattice_serial_sweep_path():
    """More than 64 nodes per layer: the sweep takes its serial form (lane = node). Small oval, 70 nodes per layer."""
    from oracle.oracle_lib import OracleBackend
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import make_oval_lattice
    lat = make_oval_lattice(num_layers=60, nodes_per_layer=70, layer_spacing=10.0, lat_resolution=0.25, lat_steps=3,
                            radius=60.0, horizon=100.0, v_straight=40.0)
    assert int(lat.nodes_in_layer.max()) == 70
    hip, orc = _capi.HipBackend(lat), OracleBackend(lat)
    scen, vels = scattered_obstacle_scenarios(lat, 128, n_obj=6, seed=4)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    compare_results(hip.plan_paths(batch), orc.plan_paths(batch), lat)
    vel = vel_inputs(lat, scen, vels, 6)
    r, vr = hip.tick_batch(batch, vel)
    o, ov = orc.tick_batch(batch, vel)
    compare_tick(r, vr, o, ov)
    b1 = _capi.PathsBatch(scen[:3], w_last_edges=[0.0, 0.5, 0.8])          # latency kernel (four waves per scenario)
    compare_results(hip.plan_paths(b1), orc.plan_paths(b1), lat)


def test_capacity_and_argument_errors(monteblanco, hip_backend):
    """Error behaviour of the C ABI: status codes + messages, never a crash or a silent fallback."""
    lat = monteblanco
    scen, _ = c2_scenarios(lat, 2, seed=3)
    # more obstacle positions than the kernel's per-scenario capacity
    big = dict(scen[0])
    big["vehicles"] = [(2.5, np.zeros((1, 2)) + k) for k in range(100)]
    with pytest.raises(_capi.BackendError, match="vehicles"):
        hip_backend.plan_paths(_capi.PathsBatch([big], w_last_edges=[0.0, 0.5, 0.8]))
    # start layer out of range
    bad = dict(scen[0]); bad["start_node"] = (lat.num_layers + 5, 0)
    with pytest.raises(_capi.BackendError, match="start_layer"):
        hip_backend.plan_paths(_capi.PathsBatch([bad], w_last_edges=[0.0, 0.5, 0.8]))
    # a start node that does not exist in its layer is not an error: no path, valid = 0
    ghost = dict(scen[0]); ghost["start_node"] = (scen[0]["start_node"][0], 200)
    res = hip_backend.plan_paths(_capi.PathsBatch([ghost], w_last_edges=[0.0, 0.5, 0.8]))
    assert int(res.valid.sum()) == 0
    # velocity seam: kappa / el_lengths length contract
    params = _capi.VelParamSet(len_veh=lat.veh_length)
    with pytest.raises(_capi.BackendError, match="el_lengths"):
        hip_backend.vel_profile(params, [{"mode": _capi.VEL_FB, "kappa": np.zeros(5), "el_lengths": np.ones(5),
                                          "loc_gg": np.ones((5, 2)) * 5.0, "v_start": 10.0, "v_end": 5.0}])
    # the backend keeps working after errors
    ok = hip_backend.plan_paths(_capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8]))
    assert int(ok.valid.sum()) >= 2


def test_object_ingestion_matches_reference_and_oracle(monteblanco, hip_backend, oracle_backend):
    """ltpl_process_objects: reference golden vectors (bit-exact on-track verdicts) and a large random set vs the oracle."""
    from test_objects_golden import check_backend_against_golden
    check_backend_against_golden(hip_backend)
    lat = monteblanco
    rng = np.random.default_rng(11)
    n = 200000
    l = rng.integers(0, lat.num_layers, n)
    f = rng.uniform(0.0, 1.0, n)[:, None]
    base = lat.refline[l] * (1 - f) + lat.refline[(l + 1) % lat.num_layers] * f
    off = rng.uniform(-2.0, 2.0, n) * np.where(rng.random(n) < 0.5, lat.track_width_right[l], lat.track_width_left[l])
    p = base + lat.normvec[l] * off[:, None]
    th, v, ln = rng.uniform(-np.pi, np.pi, n), rng.uniform(0, 80, n), rng.uniform(3, 6, n)
    a = hip_backend.process_objects(p[:, 0], p[:, 1], th, v, ln)
    b = oracle_backend.process_objects(p[:, 0], p[:, 1], th, v, ln)
    assert np.array_equal(a["on_track"], b["on_track"]) and 0.2 < a["on_track"].mean() < 0.8
    assert np.allclose(a["pred_x"], b["pred_x"], rtol=0, atol=1e-11) and np.allclose(a["pred_y"], b["pred_y"], rtol=0, atol=1e-11)
    assert np.array_equal(a["radius"], b["radius"])


@pytest.mark.gpu
def test_bench_two_ranks_strong_scaling_on_one_gpu():
    """BASELINE config C4 as a bench mode: `bench.py --gpus 2 --scaling strong --batch-total 1024` -- a FIXED batch of 1 024 scenarios block-
    partitioned over the ranks (here: two ranks sharing device 0, gloo for the barrier / max): the line says "strong", counts 1 024 scenarios per
    step whatever the rank count, every rank ran 512, and rank 0's shard is checked against the oracle."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, LTPL_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--scaling", "strong",
                        "--batch-total", "1024", "--exact-steps", "--cpu-sample", "64", "--no-extra", "--latency-ticks", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["parity_checked"] is True
    assert out["config"]["batch_total"] == 1024 and out["config"]["batch_per_gpu"] == 512
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 1024) < 1e-3 * 1024
    assert len(out["per_rank_ms_per_step"]) == 2


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """SURVEY section 8e readiness without an 8-GPU node: bench.py --gpus 2 spawns two ranks (LTPL_BENCH_SHARE_GPU=1: both on device 0, gloo
    for the barrier / max / gather) -- the N > 1 code path of the driver's scaling run end to end on real kernels: one line from rank 0,
    n_gpus = 2, the value counts both shards, per-rank times, the sharded fleet leg, parity of rank 0's shard."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, LTPL_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--batch", "4096",
                        "--exact-steps", "--cpu-sample", "64", "--fleet-planners", "512", "--fleet-ticks", "40"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["parity_checked"] is True
    assert len(out["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in out["per_rank_ms_per_step"]) and out["efficiency_vs_n1"] is None
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * 4096) < 1e-3 * 2 * 4096
    fs = out["extra"]["closed_loop_device_sharded"]
    assert fs["planners_total"] == 512 and len(fs["per_rank_planner_ticks_per_s"]) == 2 and fs["matches_recording"] is True
