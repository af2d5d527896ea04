import math
import numpy as np
from .calc_ax_poss import calc_ax_poss


def calc_vel_profile_brake(kappa, el_lengths, v_start, drag_coeff, m_veh, ggv=None, loc_gg=None, dyn_model_exp=1.0,
                           mu=None, decel_max=None):
    """
    Pure forward braking profile (tph calc_vel_profile_brake): maximum deceleration from v_start until standstill
    (remaining entries stay zero). Call sites: VpForwardBackward.py:115,247; calc_vel_profile_follow.py:152,185;
    calc_brake_emergency.py:31.
    """
    if kappa.size != el_lengths.size + 1:
        raise RuntimeError("kappa must have the length of el_lengths + 1!")
    if (ggv is not None or mu is not None) and loc_gg is not None:
        raise RuntimeError("Either ggv and optionally mu OR loc_gg must be supplied, not both (or all) of them!")
    if ggv is None and loc_gg is None:
        raise RuntimeError("Either ggv or loc_gg must be supplied!")
    if loc_gg is not None:
        if loc_gg.ndim != 2:
            raise RuntimeError("loc_gg must have two dimensions!")
        if loc_gg.shape[0] != kappa.size:
            raise RuntimeError("Length of loc_gg and kappa must be equal!")
        if loc_gg.shape[1] != 2:
            raise RuntimeError("loc_gg must consist of two columns: [ax_max, ay_max]!")
    if ggv is not None and ggv.shape[1] != 3:
        raise RuntimeError("ggv diagram must consist of the three columns [vx, ax_max, ay_max]!")
    if decel_max is not None and not decel_max < 0.0:
        raise RuntimeError("Deceleration input must be negative!")

    if mu is None:
        mu = np.ones(kappa.size)

    if ggv is not None:
        p_ggv = np.repeat(np.expand_dims(ggv, axis=0), kappa.size, axis=0)
    else:
        p_ggv = np.expand_dims(np.column_stack((np.ones(loc_gg.shape[0]) * 10.0, loc_gg)), axis=1)

    vx_profile = np.zeros(kappa.size)
    vx_profile[0] = v_start

    radii = np.abs(np.divide(1.0, kappa, out=np.full(kappa.size, np.inf), where=kappa != 0.0))

    for i in range(vx_profile.size - 1):
        ggv_mod = np.copy(p_ggv[i])
        ggv_mod[:, 1] *= -1.0
        ax_final = calc_ax_poss(vx_start=vx_profile[i], radius=radii[i], ggv=ggv_mod, mu=mu[i], mode='decel_forw',
                                dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh)

        ax_drag = -math.pow(vx_profile[i], 2) * drag_coeff / m_veh
        if decel_max is not None and ax_final < decel_max:
            if ax_drag < decel_max:
                ax_final = ax_drag
            else:
                ax_final = decel_max

        radicand = math.pow(vx_profile[i], 2) + 2 * ax_final * el_lengths[i]
        if radicand < 0.0:
            break
        vx_profile[i + 1] = math.sqrt(radicand)

    return vx_profile
