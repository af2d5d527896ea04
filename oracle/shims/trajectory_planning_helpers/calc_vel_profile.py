import math
import numpy as np
from .calc_ax_poss import calc_ax_poss
from .conv_filt import conv_filt


def calc_vel_profile(ax_max_machines, kappa, el_lengths, closed, drag_coeff, m_veh, ggv=None, loc_gg=None,
                     v_max=None, dyn_model_exp=1.0, mu=None, v_start=None, v_end=None, filt_window=None):
    """
    Forward-backward velocity profile solver (tph calc_vel_profile). Only the unclosed variant is restated: every call
    of the reference passes closed=False (VpForwardBackward.py:213-225; calc_vel_profile_follow.py:268,297).
    """
    if closed:
        raise NotImplementedError("closed=True is never used by the reference; not restated in the oracle shim")
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
    if ax_max_machines.ndim != 2 or ax_max_machines.shape[1] != 2:
        raise RuntimeError("ax_max_machines must consist of the two columns [vx, ax_max_machines]!")
    if v_max is None and ggv is None:
        raise RuntimeError("v_max must be supplied if ggv is None!")
    if kappa.size != el_lengths.size + 1:
        raise RuntimeError("kappa must have the length of el_lengths + 1 if closed is False!")
    if v_start is None:
        raise RuntimeError("v_start must be provided for the unclosed case!")
    if v_start < 0.0:
        v_start = 0.0
    if v_end is not None and v_end < 0.0:
        v_end = 0.0
    if not 1.0 <= dyn_model_exp <= 2.0:
        print("WARNING: Exponent for the vehicle dynamics model should be in the range [1.0, 2.0]!")

    if mu is None:
        mu = np.ones(kappa.size)

    if ggv is not None:
        p_ggv = np.repeat(np.expand_dims(ggv, axis=0), kappa.size, axis=0)
        op_mode = 'ggv'
    else:
        p_ggv = np.expand_dims(np.column_stack((np.ones(loc_gg.shape[0]) * 10.0, loc_gg)), axis=1)
        op_mode = 'loc_gg'

    if v_max is None:
        v_max = min(ggv[-1, 0], ax_max_machines[-1, 0])

    radii = np.abs(np.divide(1.0, kappa, out=np.full(kappa.size, np.inf), where=kappa != 0.0))

    vx_profile = _solver_fb_unclosed(p_ggv=p_ggv, ax_max_machines=ax_max_machines, v_max=v_max, radii=radii,
                                     el_lengths=el_lengths, mu=mu, v_start=v_start, v_end=v_end,
                                     dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh,
                                     op_mode=op_mode)

    if filt_window is not None:
        vx_profile = conv_filt(vx_profile, filt_window, closed)

    return vx_profile


def _solver_fb_unclosed(p_ggv, ax_max_machines, v_max, radii, el_lengths, mu, v_start, v_end, dyn_model_exp,
                        drag_coeff, m_veh, op_mode):
    if op_mode == 'ggv':
        mu_mean = np.mean(mu)
        ay_max_global = mu_mean * np.amin(p_ggv[0, :, 2])
        vx_profile = np.sqrt(ay_max_global * radii)
        ay_max_curr = mu * np.interp(vx_profile, p_ggv[0, :, 0], p_ggv[0, :, 2])
        vx_profile = np.sqrt(np.multiply(ay_max_curr, radii))
    else:
        ay_max_curr = mu * p_ggv[:, 0, 2]
        vx_profile = np.sqrt(ay_max_curr * radii)

    vx_profile[vx_profile > v_max] = v_max

    if vx_profile[0] > v_start:
        vx_profile[0] = v_start

    vx_profile = _solver_fb_acc_profile(p_ggv=p_ggv, ax_max_machines=ax_max_machines, v_max=v_max, radii=radii,
                                        el_lengths=el_lengths, mu=mu, vx_profile=vx_profile,
                                        dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh,
                                        backwards=False)

    if v_end is not None and vx_profile[-1] > v_end:
        vx_profile[-1] = v_end

    vx_profile = _solver_fb_acc_profile(p_ggv=p_ggv, ax_max_machines=ax_max_machines, v_max=v_max, radii=radii,
                                        el_lengths=el_lengths, mu=mu, vx_profile=vx_profile,
                                        dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh,
                                        backwards=True)
    return vx_profile


def _solver_fb_acc_profile(p_ggv, ax_max_machines, v_max, radii, el_lengths, mu, vx_profile, dyn_model_exp,
                           drag_coeff, m_veh, backwards=False):
    """
    One sweep over the profile. NOTE (restated quirk): in the backward sweep radii, element lengths, mu and the profile
    are flipped, but the per-point gg table ``p_ggv`` is indexed UNflipped. With a constant local gg (the only case the
    stock examples exercise: local_gg=(5.0, 5.0), Graph_LTPL.py:349) both readings coincide.
    """
    no_points = vx_profile.size

    if backwards:
        radii_mod = np.flipud(radii)
        el_lengths_mod = np.flipud(el_lengths)
        mu_mod = np.flipud(mu)
        vx_profile = np.flipud(vx_profile)
        mode = 'decel_backw'
    else:
        radii_mod = radii
        el_lengths_mod = el_lengths
        mu_mod = mu
        mode = 'accel_forw'

    vx_diffs = np.diff(vx_profile)
    acc_inds = np.where(vx_diffs > 0.0)[0]
    if acc_inds.size != 0:
        acc_inds_diffs = np.diff(acc_inds)
        acc_inds_diffs = np.insert(acc_inds_diffs, 0, 2)
        acc_inds_rel = list(acc_inds[acc_inds_diffs > 1])
    else:
        acc_inds_rel = []

    while acc_inds_rel:
        i = acc_inds_rel.pop(0)

        while i < no_points - 1:
            ax_possible_cur = calc_ax_poss(vx_start=vx_profile[i], radius=radii_mod[i], ggv=p_ggv[i],
                                           ax_max_machines=ax_max_machines, mu=mu_mod[i], mode=mode,
                                           dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh)

            vx_possible_next = math.sqrt(math.pow(vx_profile[i], 2) + 2 * ax_possible_cur * el_lengths_mod[i])

            if backwards:
                for j in range(1):
                    ax_possible_next = calc_ax_poss(vx_start=vx_possible_next, radius=radii_mod[i + 1],
                                                    ggv=p_ggv[i + 1], ax_max_machines=ax_max_machines,
                                                    mu=mu_mod[i + 1], mode=mode, dyn_model_exp=dyn_model_exp,
                                                    drag_coeff=drag_coeff, m_veh=m_veh)
                    vx_tmp = math.sqrt(math.pow(vx_profile[i], 2) + 2 * ax_possible_next * el_lengths_mod[i])
                    if vx_tmp < vx_possible_next:
                        vx_possible_next = vx_tmp
                    else:
                        break

            if vx_possible_next < vx_profile[i + 1]:
                vx_profile[i + 1] = vx_possible_next

            i += 1

            if vx_possible_next > v_max or (acc_inds_rel and i >= acc_inds_rel[0]):
                break

    if backwards:
        vx_profile = np.flipud(vx_profile)

    return vx_profile
