"""
Host-side mirror of seam (2): class ``VpForwardBackward`` (graph_ltpl/online_graph/src/VpForwardBackward.py:11-255) with
the same constructor, methods, argument names and return values; the arithmetic (tph.calc_vel_profile,
tph.calc_vel_profile_brake and calc_vel_profile_follow.py) runs in the velocity kernels of csrc/ltpl_hip.hip.
"""
import logging
import numpy as np

from . import _capi


class VpForwardBackward(object):
    backend = None          # set by install() / tests: a _capi.HipBackend bound to the planner's lattice

    def __init__(self, dyn_model_exp: float, drag_coeff: float, m_veh: float, len_veh: float,
                 follow_control_type: str, follow_control_params: dict, glob_rl: np.ndarray, backend=None) -> None:
        self.__log = logging.getLogger("local_trajectory_logger")
        if backend is not None:
            self.backend = backend
        if self.backend is None:
            raise _capi.BackendError("VpForwardBackward: no HIP backend bound (call install() first); "
                                     "there is no CPU fallback")
        self.__follow_control_type = follow_control_type
        self.__follow_control_params = follow_control_params
        self.__dyn_model_exp = dyn_model_exp
        self.__drag_coeff = drag_coeff
        self.__m_veh = m_veh
        self.__len_veh = len_veh
        # the global race line already lives on the device (ltpl_lattice_desc.glob_rl); keep the reference for checks
        self.__glob_rl_clsd = glob_rl
        self.__vel_max = None
        self.__gg_scale = None
        self.__ax_max_machines = None
        self.__old_gg_scale = None
        self.__params = None

    def update_dyn_parameters(self, vel_max: float, gg_scale: float, ax_max_machines: np.ndarray) -> None:
        """VpForwardBackward.py:65-84"""
        if self.__old_gg_scale is None:
            self.__old_gg_scale = gg_scale
        self.__vel_max = vel_max
        self.__gg_scale = gg_scale
        self.__ax_max_machines = ax_max_machines
        self.__params = _capi.VelParamSet(dyn_model_exp=self.__dyn_model_exp, drag_coeff=self.__drag_coeff,
                                          m_veh=self.__m_veh, len_veh=self.__len_veh, v_max=vel_max,
                                          ax_max_machines=ax_max_machines,
                                          follow_control_type=self.__follow_control_type,
                                          follow_control_params=self.__follow_control_params)

    def __brake(self, loc_gg, kappa, el_lengths, v_start):
        if kappa.size != el_lengths.size + 1:
            raise RuntimeError("kappa must have the length of el_lengths + 1!")
        job = {"mode": _capi.VEL_BRAKE, "kappa": kappa, "el_lengths": el_lengths, "loc_gg": loc_gg,
               "v_start": v_start}
        return self.backend.vel_profile(self.__params, [job])[0][0]

    def check_brake_prefix(self, vel_plan: float, vel_course: np.ndarray, kappa: np.ndarray, el_lengths: np.ndarray,
                           loc_gg: np.ndarray) -> tuple:
        """VpForwardBackward.py:86-139: (velocity prefix, number of prefix points behind vel_course, start velocity of the profile).
        Only when the planned velocity exceeds the (lowered) maximum: brake with the friction scale of the PREVIOUS ticks until the
        maximum is met; that scale only advances while no such braking phase is pending."""
        if not vel_plan > self.__vel_max + 0.1:
            self.__old_gg_scale = self.__gg_scale
            return vel_course, 0, vel_plan
        self.__log.info("Applying deceleration in order to break to new v_max!")
        decel = self.__brake(loc_gg * self.__old_gg_scale, kappa, el_lengths, vel_plan)
        below = np.flatnonzero(decel <= self.__vel_max)
        cut = int(below[0]) if below.size and below[0] > 0 else len(decel) - 1      # np.argmax == 0 -> last point (:124-126)
        return np.concatenate((vel_course, decel[:cut])), cut, decel[cut]

    def calc_vel_profile_follow(self, kappa: np.ndarray, el_lengths: np.ndarray, loc_gg: np.array, v_start: float,
                                v_ego: float, v_obj: float, safety_d: float, obj_dist: float,
                                obj_pos: list) -> tuple:
        """VpForwardBackward.py:141-192 -> calc_vel_profile_follow.py:78-313"""
        job = {"mode": _capi.VEL_FOLLOW, "kappa": kappa, "el_lengths": el_lengths,
               "loc_gg": loc_gg * self.__gg_scale, "v_start": v_start, "v_ego": v_ego, "v_obj": v_obj,
               "safety_d": safety_d, "obj_dist": obj_dist, "obj_pos": obj_pos}
        vx, too_close, vel_bound = self.backend.vel_profile(self.__params, [job])[0]
        return vx, too_close, vel_bound

    def calc_vel_profile(self, kappa: np.ndarray, el_lengths: np.ndarray, loc_gg: np.ndarray, v_start: float,
                         v_end: float) -> np.ndarray:
        """VpForwardBackward.py:194-227"""
        if kappa.size != el_lengths.size + 1:
            raise RuntimeError("kappa must have the length of el_lengths + 1 if closed is False!")
        job = {"mode": _capi.VEL_FB, "kappa": kappa, "el_lengths": el_lengths, "loc_gg": loc_gg * self.__gg_scale,
               "v_start": v_start, "v_end": v_end}
        return self.backend.vel_profile(self.__params, [job])[0][0]

    def calc_vel_brake_em(self, kappa: np.ndarray, el_lengths: np.ndarray, loc_gg: np.array, v_start: float):
        """VpForwardBackward.py:229-255 (no gg-scale)"""
        return self.__brake(loc_gg, kappa, el_lengths, v_start)
