"""
SURVEY.md section 8f rank 4: the tick log in the reference's on-disk format (Logging.log_onlinegraph rows).

  * CPU: a closed loop of the planner (host-logic harness) writes a log through TickLogWriter; read_log returns the values that
    were written; revalidate re-plans every logged tick in one batch and finds the logged node lists.
  * container only: a log WRITTEN BY THE REFERENCE's own Logging class is parsed by read_log, and a log written by
    TickLogWriter is parsed by the reference viewer's own line parser (visualize_graph_log.get_data_from_line).
"""
import os
import warnings

import numpy as np
import pytest

import planner_replay as pr
from graphbasedlocaltrajectoryplanner_amd import tick_log
from oracle import ref_env


def write_planner_log(path, backend, planner, lat, ticks, n):
    st = ticks[0]['start']
    planner.set_start(0, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    w = tick_log.TickLogWriter(path, graph_id="test-graph")
    trajs = []
    for t in ticks[:n]:
        veh = pr.vehicles_of_tick(t)
        planner.calc_paths([t['action_id_sel']], [t['t']], [veh], [pr.zone_gids_of_tick(lat, t)])
        snap = w.snapshot_paths(planner)
        va = t['vel_args']
        planner.calc_vel_profile([t['pos_est']], va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'],
                                 local_gg=tuple(va['local_gg']), ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'])
        w.write_planner_tick(planner, backend, t['t'], t['pos_est'], t['action_id_sel'], veh,
                             t.get('zone_layers', ()), t.get('zone_nodes', ()), export_rows=115, paths_snapshot=snap)
        trajs.append(planner.trajectories(0)[0])
    return trajs


def test_planner_log_round_trip_and_revalidation(tmp_path, monteblanco, oracle_backend):
    from oracle.planner_host import HostPlannerBackend
    ticks = pr.load_ticks("c2")
    planner = HostPlannerBackend(monteblanco).planner(1)
    path = str(tmp_path / "ticks_data.csv")
    trajs = write_planner_log(path, oracle_backend, planner, monteblanco, ticks, 150)
    graph_id, rows = tick_log.read_log(path)
    assert graph_id == "test-graph" and len(rows) == 150
    for r, t, tr in zip(rows, ticks, trajs):
        assert r["time"] == t["t"] and r["action_id_prev"] == t["action_id_sel"]
        assert r["start_node"] == t["paths"]["start_node"]
        assert list(r["vel_list"].keys()) == t["vel"]["keys"]
        assert {k: v[0] for k, v in r["nodes_list"].items()} == t["paths"]["nodes"]
        for k in r["vel_list"]:
            assert np.array_equal(np.array(r["vel_list"][k][0]), tr[k][0][:115, 5])      # repr round trip is exact
            assert np.array_equal(np.array(r["pos_list"][k][0]), tr[k][0][:115, 1:3])
        assert len(r["obj_veh"]) == len(t["obj_radius"])
    assert rows[5]["obj_zone"] == rows[0]["obj_zone"] and rows[0]["obj_zone"]        # "no update since" resolved
    bad = tick_log.revalidate(oracle_backend, monteblanco, rows, w_last_edges=())
    # re-planning without the constant segment and without the previous-solution discount (like the stock viewer) restores the
    # logged node lists on (nearly) every tick
    assert len(bad) <= len(rows) // 10, bad
    # with the context the neighbouring rows hold (previous solution, position estimate, constant segment): EVERY tick, every key
    assert tick_log.revalidate(oracle_backend, monteblanco, rows, w_last_edges=(0.0, 0.5, 0.8), context=True) == []
    # the constant segment is logged from the current position on (Graph_LTPL.py:445-447): its last point is the start node
    for r in rows[1:]:
        if r["const_path_seg"]:
            l, n = r["start_node"]
            assert np.allclose(r["const_path_seg"][-1], monteblanco.node_pos[monteblanco.layer_off[l] + n], atol=1e-9)


@pytest.mark.gpu
def test_revalidation_on_the_device(tmp_path, monteblanco, hip_backend, oracle_backend):
    """SURVEY.md section 8f rank 4 on the MI355X: a closed loop of the HIP planner writes the log, the batched re-validation runs on the
    HIP backend (one launch for all rows). Strict: with the rows' context every logged node list is reproduced; in the stock
    viewer's context-free form the device finds exactly the mismatches the oracle finds."""
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    ticks = pr.load_ticks("c2")
    planner = Planner(hip_backend, 1)
    path = str(tmp_path / "ticks_data.csv")
    write_planner_log(path, hip_backend, planner, monteblanco, ticks, 400)
    planner.close()
    _, rows = tick_log.read_log(path)
    assert len(rows) == 400
    for r, t in zip(rows, ticks):
        assert {k: v[0] for k, v in r["nodes_list"].items()} == t["paths"]["nodes"]
    assert tick_log.revalidate(hip_backend, monteblanco, rows, w_last_edges=(0.0, 0.5, 0.8), context=True) == []
    assert tick_log.revalidate(hip_backend, monteblanco, rows) == tick_log.revalidate(oracle_backend, monteblanco, rows)


@pytest.mark.reference
@pytest.mark.skipif(not ref_env.reference_available(), reason="reference tree not present")
def test_format_is_interchangeable_with_the_reference(tmp_path, monteblanco, oracle_backend):
    from oracle import ref_scenarios as rs
    warnings.simplefilter("ignore")
    # (a) a log written by the reference's own Logging class
    gl, clock = ref_env.load_reference()
    path_dict = ref_env.default_path_dict(os.path.join(pr.__file__.rsplit("/", 2)[0], "oracle", "_cache"))
    path_dict.update({'log_path': str(tmp_path) + "/logs/", 'graph_log_id': "reftest"})
    ltpl_obj = gl.Graph_LTPL.Graph_LTPL(path_dict=path_dict, visual_mode=False, log_to_file=True)
    ltpl_obj.graph_init()
    exported = rs.run_loop(gl, clock, ltpl_obj, path_dict, n_ticks=40, dt=0.05, dummies=rs.opponents_c2(gl, 8),
                           zones=rs.ZONE_EXAMPLE, on_tick=lambda i, e: ltpl_obj.log())
    import logging
    for h in list(logging.getLogger("local_trajectory_logger").handlers):
        if isinstance(h, logging.FileHandler):
            logging.getLogger("local_trajectory_logger").removeHandler(h)
    graph_id, rows = tick_log.read_log(path_dict['graph_log_data_path'])
    assert len(rows) == 40
    for r, e in zip(rows, exported):
        assert list(r["vel_list"].keys()) == list(e["traj"].keys())
        for k in e["traj"]:
            assert np.allclose(np.array(r["vel_list"][k][0]), e["traj"][k][:, 5], rtol=0, atol=1e-12)
    assert not tick_log.revalidate(oracle_backend, monteblanco, rows[:1])
    # (b) a log written by TickLogWriter, parsed by the stock viewer's line parser
    from oracle.planner_host import HostPlannerBackend
    planner = HostPlannerBackend(monteblanco).planner(1)
    mine = str(tmp_path / "mine_data.csv")
    write_planner_log(mine, oracle_backend, planner, monteblanco, pr.load_ticks("c2"), 30)
    src = open(os.path.join(ref_env.REFERENCE_ROOT, "graph_ltpl", "visualization", "src", "visualize_graph_log.py")).read()
    start = src.index("def get_data_from_line(")
    end = src.index("\n\n\n", start)
    ns = {"json": __import__("json"), "np": np}
    exec(src[start:end], ns)                                   # the reference's parser, verbatim, without its GUI imports
    out = ns["get_data_from_line"](mine, 12)
    assert out[0] == tick_log.read_log(mine)[1][12]["start_node"]      # start node decoded identically by both parsers
    assert isinstance(out[4], dict) and isinstance(out[6], dict)
