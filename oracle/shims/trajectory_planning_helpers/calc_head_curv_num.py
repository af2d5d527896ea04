import math
import numpy as np
from .normalize_psi import normalize_psi


def calc_head_curv_num(path, el_lengths, is_closed, stepsize_psi_preview=1.0, stepsize_psi_review=1.0,
                       stepsize_curv_preview=2.0, stepsize_curv_review=2.0, calc_curv=True):
    """
    Numerical heading / curvature by finite differences over preview / review index windows (tph). The reference only
    uses it offline and only with is_closed=True (gen_node_skeleton.py:63,84,88; objectlist_dummy.py:115), keeping [0].
    """
    if is_closed and path.shape[0] != el_lengths.size:
        raise RuntimeError("path and el_lenghts must have the same length!")
    if not is_closed and path.shape[0] != el_lengths.size + 1:
        raise RuntimeError("path must have the length of el_lengths + 1!")
    if not is_closed:
        raise NotImplementedError("unclosed variant is never reached by the reference; not restated")

    no_points = path.shape[0]
    avg = float(np.average(el_lengths))
    ind_step_preview_psi = max(int(round(stepsize_psi_preview / avg)), 1)
    ind_step_review_psi = max(int(round(stepsize_psi_review / avg)), 1)
    ind_step_preview_curv = max(int(round(stepsize_curv_preview / avg)), 1)
    ind_step_review_curv = max(int(round(stepsize_curv_review / avg)), 1)
    steps_tot_psi = ind_step_preview_psi + ind_step_review_psi
    steps_tot_curv = ind_step_preview_curv + ind_step_review_curv

    path_temp = np.vstack((path[-ind_step_review_psi:], path, path[:ind_step_preview_psi]))
    tangvecs = np.stack((path_temp[steps_tot_psi:, 0] - path_temp[:-steps_tot_psi, 0],
                         path_temp[steps_tot_psi:, 1] - path_temp[:-steps_tot_psi, 1]), axis=1)
    psi = np.arctan2(tangvecs[:, 1], tangvecs[:, 0]) - math.pi / 2
    psi = normalize_psi(psi)

    if calc_curv:
        psi_temp = np.insert(psi, 0, psi[-ind_step_review_curv:])
        psi_temp = np.append(psi_temp, psi[:ind_step_preview_curv])
        delta_psi = np.zeros(no_points)
        for i in range(no_points):
            delta_psi[i] = normalize_psi(psi_temp[i + steps_tot_curv] - psi_temp[i])
        s_points_cl = np.cumsum(el_lengths)
        s_points_cl = np.insert(s_points_cl, 0, 0.0)
        s_points = s_points_cl[:-1]
        s_points_cl_reverse = np.flipud(-np.cumsum(np.flipud(el_lengths)))
        s_points_temp = np.insert(s_points, 0, s_points_cl_reverse[-ind_step_review_curv:])
        s_points_temp = np.append(s_points_temp, s_points_cl[-1] + s_points[:ind_step_preview_curv])
        kappa = delta_psi / (s_points_temp[steps_tot_curv:] - s_points_temp[:-steps_tot_curv])
    else:
        kappa = 0.0

    return psi, kappa
