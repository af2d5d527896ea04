"""
Drop-in for the reference's class ``OnlineTrajectoryHandler`` (graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:24-1040,
constructed at Graph_LTPL.py:221-227) on top of the planner entry points of the C ABI (``ltpl_planner_*``): the iterative
memory lives in C++ behind the ABI, every tick stage is one C call, and this class only converts between the reference's Python
objects (VehObject / ZoneObject instances in, dicts of one-element lists out) and flat buffers.

Same constructor arguments, public methods and return tuples as the reference class: set_initial_pose, update_objects,
calc_paths, get_ref_idx, calc_vel_profile.
"""
import configparser
import json
import time

import numpy as np

from .planner import Planner
from .zones import ZoneFilter


def _read_ini(path):
    cfg = configparser.ConfigParser()
    if not cfg.read(path):
        raise ValueError('Specified graph config file does not exist or is empty!')
    return cfg


def planner_config_from_ini(online_ini, dyn_model_exp, drag_coeff, m_veh) -> dict:
    """Keyword arguments of ``Planner`` from params/ltpl_config_online.ini (the keys OTH.__init__ reads, OTH.py:99-122)."""
    if online_ini.get('VP', 'vp_type') != "fb":
        raise ValueError("only the forward-backward velocity planner (vp_type=fb) is available in this backend")
    ctrl = online_ini.get('FOLLOW', 'controller_type')
    return dict(w_last_edges=json.loads(online_ini.get('COST', 'w_last_edges')),
                v_max_offset=online_ini.getfloat('ACTIONSET', 'v_max_offset'),
                delaycomp=online_ini.getfloat('DELAY', 'delaycomp'),
                calc_time_safety=online_ini.getfloat('CALC_TIME', 'calc_time_safety'),
                calc_time_buffer_len=online_ini.getint('CALC_TIME', 'calc_time_buffer_len'),
                filt_window_width=online_ini.getint('SMOOTHING', 'filt_window_width'),
                dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh, follow_control_type=ctrl,
                follow_control_params=json.loads(online_ini.get('FOLLOW', 'control_params_' + ctrl)))


class PlannerOnlineTrajectoryHandler(object):
    session = None          # set by install(): owns the lattice upload per GraphBase and (tests) the clock

    def __init__(self, graph_base, graph_online_config_path: str, graph_offline_config_path: str,
                 veh_param_dyn_model_exp: float = 1.0, veh_param_dragcoeff: float = 0.85,
                 veh_param_mass: float = 1000.0) -> None:
        online = _read_ini(graph_online_config_path)
        _read_ini(graph_offline_config_path)
        _, self._lat, self._backend, _ = self.session.bind(graph_base)
        cfg = planner_config_from_ini(online, veh_param_dyn_model_exp, veh_param_dragcoeff, veh_param_mass)
        make = getattr(self._backend, "planner", None)            # test harness backends bring their own constructor
        self._planner = make(1, **cfg) if make is not None else Planner(self._backend, 1, **cfg)
        self._graph_base = graph_base
        self._zones = ZoneFilter(self._lat)
        self._obj_veh, self._obj_zone = [], []
        self._clock = self.session.clock if self.session.clock is not None else time
        self._pos_est = None

    # ---- OTH.py:181-270 -----------------------------------------------------------------------------------------------
    def set_initial_pose(self, start_pos: list, start_heading: float, start_vel: float = 0.0,
                         max_heading_offset: float = np.pi / 4) -> tuple:
        return self._planner.set_start(0, start_pos, float(np.squeeze(start_heading)), start_vel, max_heading_offset)

    # ---- OTH.py:272-287 -----------------------------------------------------------------------------------------------
    def update_objects(self, obj_veh: list, obj_zone: list) -> None:
        self._obj_veh, self._obj_zone = obj_veh, obj_zone

    # ---- OTH.py:289-516 -----------------------------------------------------------------------------------------------
    def calc_paths(self, action_id_sel: str, idx_sel_traj: int) -> tuple:
        if idx_sel_traj != 0:
            raise ValueError("one trajectory per action set (max_solutions has no effect in the reference either)")
        vehicles = []
        for v in self._obj_veh:
            pos = np.asarray(v.get_pos(), dtype=np.float64).reshape(1, 2)
            pred = v.get_prediction()
            pred = np.zeros((0, 2)) if pred is None else np.asarray(pred, dtype=np.float64).reshape(-1, 2)
            vehicles.append((float(v.get_radius()), float(v.get_vel()), np.vstack((pos, pred))))
        pl = self._planner
        pl.calc_paths_begin([action_id_sel], [self._clock.time()], [vehicles])
        # the zone filter is decided on the start node of THIS search (gen_local_node_template.py:42-99)
        self._zones.refresh(self._graph_base, pl.start_node(0)[0], self._obj_zone)
        pl.calc_paths_finish([self._zones.gids])
        p = pl.paths(0)
        path_dict = {k: [p["path_param"][k]] for k in p["keys"]}
        node_dict = {k: [p["nodes"][k]] for k in p["keys"]}
        const_seg = None
        if p["const_rows"] >= 0 and p["keys"]:
            const_seg = p["path_param"][p["keys"][0]][:p["const_rows"], :]
        return path_dict, p["start_node"], node_dict, const_seg

    # ---- OTH.py:518-601 -----------------------------------------------------------------------------------------------
    def get_ref_idx(self, action_id_sel: str, idx_sel_traj: int, pos_est: tuple) -> tuple:
        self._pos_est = [float(pos_est[0]), float(pos_est[1])]
        return self._planner.get_ref_idx([self._pos_est])

    # ---- OTH.py:603-1040 ----------------------------------------------------------------------------------------------
    def calc_vel_profile(self, cut_index_pos: int, cut_layer: int, vel_plan: float, acc_plan: float,
                         vel_course: np.ndarray, vel_est: float, vel_max: float, ax_max_machines: np.ndarray,
                         safety_d: float, gg_scale: float, local_gg: dict = (5.0, 5.0),
                         incl_emerg_traj: bool = False) -> tuple:
        if not isinstance(local_gg, dict) and (type(local_gg) is not tuple or len(local_gg) != 2):
            raise ValueError("Provided local_gg does not satisfy requested format! Read parameter documentation.")
        stamp = self._clock.time()
        # the reference index of this tick already lives in the planner (get_ref_idx above); the pose is only re-sent
        self._planner.calc_vel_profile([self._pos_est], vel_est, vel_max=vel_max, gg_scale=gg_scale, local_gg=local_gg,
                                       ax_max_machines=ax_max_machines, safety_d=safety_d, incl_emerg_traj=incl_emerg_traj)
        action_set, ids, _ = self._planner.trajectories(0)
        return action_set, ids, stamp, [t[0][:, 1:3] for t in action_set.values()]
