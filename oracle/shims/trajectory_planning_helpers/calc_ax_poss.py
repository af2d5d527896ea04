import math
import numpy as np


def calc_ax_poss(vx_start, radius, ggv, mu, dyn_model_exp, drag_coeff, m_veh, ax_max_machines=None,
                 mode='accel_forw'):
    """
    Longitudinal acceleration still available at speed ``vx_start`` on curve radius ``radius`` (tph calc_vel_profile
    helper): tire share from the (generalised) friction ellipse with exponent ``dyn_model_exp``, capped by the machine
    limit when accelerating forward, then drag added (forward modes) or subtracted (backward deceleration).
    """
    if mode not in ['accel_forw', 'decel_forw', 'decel_backw']:
        raise RuntimeError("Unknown operation mode for calc_ax_poss!")
    if mode == 'accel_forw' and ax_max_machines is None:
        raise RuntimeError("ax_max_machines is required if operation mode is accel_forw!")
    if ggv.ndim != 2 or ggv.shape[1] != 3:
        raise RuntimeError("ggv must have two dimensions and three columns [vx, ax_max, ay_max]!")

    ax_max_tires = mu * np.interp(vx_start, ggv[:, 0], ggv[:, 1])
    ay_max_tires = mu * np.interp(vx_start, ggv[:, 0], ggv[:, 2])
    ay_used = math.pow(vx_start, 2) / radius

    if mode in ['accel_forw', 'decel_backw'] and ax_max_tires < 0.0:
        ax_max_tires *= -1.0
    elif mode == 'decel_forw' and ax_max_tires > 0.0:
        ax_max_tires *= -1.0

    radicand = 1.0 - math.pow(ay_used / ay_max_tires, dyn_model_exp)
    if radicand > 0.0:
        ax_avail_tires = ax_max_tires * math.pow(radicand, 1.0 / dyn_model_exp)
    else:
        ax_avail_tires = 0.0

    if mode == 'accel_forw':
        ax_max_machines_tmp = np.interp(vx_start, ax_max_machines[:, 0], ax_max_machines[:, 1])
        ax_avail_vehicle = min(ax_avail_tires, ax_max_machines_tmp)
    else:
        ax_avail_vehicle = ax_avail_tires

    ax_drag = -math.pow(vx_start, 2) * drag_coeff / m_veh

    if mode in ['accel_forw', 'decel_forw']:
        ax_final = ax_avail_vehicle + ax_drag
    else:
        ax_final = ax_avail_vehicle - ax_drag

    return ax_final
