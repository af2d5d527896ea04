"""C5 latency run: high-resolution oval (0.5 m layer spacing, 21 lateral nodes; horizon in metres = first argument,
default 300 = 600 layers = the long-horizon mode, 100 = the fused LDS-resident kernel), a slow opponent ahead so that the follow-mode velocity profile runs on every tick; single-scenario synchronous
ltpl_tick_batch calls, host wall time including marshalling and PCIe."""
import json
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphbasedlocaltrajectoryplanner_amd import _capi                                   # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import raceline_state               # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c5_lattice            # noqa: E402

HORIZON = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
N_TICKS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
lat = c5_lattice(horizon=HORIZON)
hip = _capi.HipBackend(lat)
rng = np.random.default_rng(2)
singles = []
params = _capi.VelParamSet(len_veh=lat.veh_length)
for _ in range(64):
    sl = int(rng.integers(0, lat.num_layers)); sn = int(lat.raceline_index[sl])
    x, y, psi, v = raceline_state(lat, float(lat.s_raceline[sl]) + rng.uniform(20.0, 80.0))
    v = float(v) * rng.uniform(0.2, 0.5)
    pred = np.array([[x - np.sin(psi) * v * 0.2, y + np.cos(psi) * v * 0.2]])
    sc = {"start_node": (sl, sn), "action_sets": True, "vehicles": [(2.5, np.vstack((np.array([[x, y]]), pred)))], "zone_gids": [],
          "last_nodes": None, "obj_in_const": False, "obj_besides": False, "last_action": None, "const_closest": None,
          "psi_s": float(lat.node_psi[lat.layer_off[sl] + sn])}
    b1 = _capi.PathsBatch([sc], w_last_edges=[0.0, 0.5, 0.8])
    vp = float(rng.uniform(5.0, 35.0))
    v1 = _capi.TickVelBatch(params, 1, [vp], [vp], lat.node_pos[lat.layer_off[sl] + sn][None, :], np.array([v]))
    singles.append((b1, v1))
res, vres = hip.new_paths_result(1), _capi.TickVelResult(1, hip.caps.max_path_pts)
lat_us, n_follow = [], 0
for i in range(100 + N_TICKS):
    b1, v1 = singles[i % 64]
    t1 = time.perf_counter()
    hip.tick_batch(b1, v1, res, vres)
    if i >= 100:
        lat_us.append((time.perf_counter() - t1) * 1e6)
        n_follow += int(((res.action_id == _capi.ACT_FOLLOW) & (res.valid == 1)).sum())
lat_us = np.array(lat_us)
print(json.dumps({"config": "C5: high-res oval, %d layers x %d nodes, %d edges, horizon %d layers, %d path samples, follow profile on "
                            "%.0f %% of the ticks" % (lat.num_layers, int(lat.nodes_in_layer.max()), lat.num_edges,
                                                      hip.caps.max_path_nodes, hip.caps.max_path_pts, 100.0 * n_follow / lat_us.size),
                  "latency_us": {"p50": float(np.percentile(lat_us, 50)), "p99": float(np.percentile(lat_us, 99)),
                                 "mean": float(lat_us.mean()), "ticks": int(lat_us.size)}}))
