"""
GPU: the COMPOSITION the drop-in consists of -- ``oth_adapter.PlannerOnlineTrajectoryHandler`` (the class ``install(mode="planner")``
puts in place of the reference's OnlineTrajectoryHandler, Graph_LTPL.py:221-227) on top of ``Planner`` on top of ``HipBackend`` --
driven through the adapter's reference-shaped methods (update_objects / calc_paths / get_ref_idx / calc_vel_profile with
VehObject-like and ZoneObject-like arguments, the session's clock) with the recorded inputs of the unmodified reference, tick by
tick. /root/reference does not exist on the GPU box, so the facade itself cannot run there (tests/test_dropin_reference.py runs it in
the build container on the oracle's arithmetic); this test closes the other half: everything BELOW the facade on the real kernels.
"""
import os

import numpy as np
import pytest

import planner_replay as pr
from helpers import Veh
from graphbasedlocaltrajectoryplanner_amd import oth_adapter

pytestmark = pytest.mark.gpu

ONLINE_INI = """
[VP]
vp_type=fb
[COST]
w_last_edges=[0.0, 0.5, 0.8]
[ACTIONSET]
v_max_offset=0.1
[DELAY]
delaycomp=0.1
[CALC_TIME]
calc_time_safety=2.0
calc_time_buffer_len=5
[SMOOTHING]
filt_window_width=1
[FOLLOW]
controller_type=PD
control_params_PD={"c_p": 1.25, "k_d": 0.025, "k_p": 0.2}
"""


class Clock(object):
    now = 0.0

    def time(self):
        return self.now


class Zone(object):
    """Duck-typed ZoneObject (ObjectListInterface.py:299-379) carrying the recorded node list."""

    def __init__(self, layers, nodes):
        self._l, self._n, self.processed, self.disabled, self.fixed = list(layers), list(nodes), False, False, True

    def get_blocked_nodes(self, graph_base):
        return self._l, self._n

    def set_processed(self):
        self.processed = True


class Session(object):
    def __init__(self, lat, backend, clock):
        self._ent, self.clock = (None, lat, backend, None), clock

    def bind(self, graph_base):
        return self._ent


class AdapterAsPlanner(object):
    """Lets tests/planner_replay.replay drive the ADAPTER: inputs go through its reference-shaped methods, the state it leaves is read
    back through the planner it wraps -- and what the adapter hands to the facade is compared with that state on the way."""

    n_scen = 1

    def __init__(self, oth, clock):
        self.oth, self.clock, self._zone, self._zsig = oth, clock, [], None

    def set_start(self, scen, pos, heading, vel, max_heading_offset):
        return self.oth.set_initial_pose(list(pos), heading, vel, max_heading_offset)

    def calc_paths(self, prev_actions, t_now, vehicles, zone_gids):
        self.clock.now = float(t_now[0])
        veh = [Veh(pos[0], r, np.asarray(pos[1:]).reshape(-1, 2), v) for r, v, pos in vehicles[0]]
        lat = self.oth._lat
        gids = [int(g) for g in zone_gids[0]]
        if tuple(gids) != self._zsig:                       # a zone object arrives once; afterwards it is "processed"
            layers = [int(np.searchsorted(lat.layer_off, g, side="right") - 1) for g in gids]
            self._zone = [Zone(layers, [g - int(lat.layer_off[l]) for g, l in zip(gids, layers)])] if gids else []
            self._zsig = tuple(gids)
        self.oth.update_objects(veh, self._zone)
        path_dict, start_node, node_dict, const_seg = self.oth.calc_paths(prev_actions[0], 0)
        p = self.oth._planner.paths(0)
        assert list(path_dict.keys()) == p["keys"] and list(start_node) == list(p["start_node"])
        for k in p["keys"]:
            assert np.array_equal(path_dict[k][0], p["path_param"][k]) and node_dict[k][0] == p["nodes"][k]
        assert (const_seg is None) == (p["const_rows"] < 0 or not p["keys"])
        self._last_prev = prev_actions[0]

    def paths(self, scen=0):
        return self.oth._planner.paths(0)

    def calc_vel_profile(self, pos_est, vel_est, vel_max, gg_scale, local_gg, ax_max_machines, safety_d, incl_emerg_traj):
        ref = self.oth.get_ref_idx(self._last_prev, 0, tuple(pos_est[0]))
        lgg = local_gg[0] if isinstance(local_gg, list) else local_gg
        out = self.oth.calc_vel_profile(cut_index_pos=ref[0], cut_layer=ref[1], vel_plan=ref[2], acc_plan=ref[4], vel_course=ref[3],
                                        vel_est=vel_est, vel_max=vel_max, ax_max_machines=np.asarray(ax_max_machines), safety_d=safety_d,
                                        gg_scale=gg_scale, local_gg=lgg, incl_emerg_traj=incl_emerg_traj)
        self._out = out

    def trajectories(self, scen=0):
        action_set, ids, ref = self.oth._planner.trajectories(0)
        a2, ids2, stamp, coords = self._out
        assert list(a2.keys()) == list(action_set.keys()) and ids2 == ids and stamp == self.clock.now
        for k in action_set:
            assert np.array_equal(a2[k][0], action_set[k][0])
        assert len(coords) == len(action_set)
        return action_set, ids, ref


@pytest.mark.parametrize("name", ["c2", "ggmap"])
def test_adapter_on_hip_matches_reference_recordings(tmp_path, hip_backend, monteblanco, name):
    ini = tmp_path / "online.ini"
    ini.write_text(ONLINE_INI)
    clock = Clock()

    class Bound(oth_adapter.PlannerOnlineTrajectoryHandler):
        session = Session(monteblanco, hip_backend, clock)

    oth = Bound(graph_base=object(), graph_online_config_path=str(ini), graph_offline_config_path=str(ini))
    ticks = pr.load_ticks(name)[:600]
    seen = pr.replay(AdapterAsPlanner(oth, clock), monteblanco, ticks)
    assert seen['full'] >= 5 and "follow" in seen['keys']
    oth._planner.close()
