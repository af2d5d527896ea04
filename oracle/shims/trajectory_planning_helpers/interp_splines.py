import math
import numpy as np
from .calc_spline_lengths import calc_spline_lengths


def interp_splines(coeffs_x, coeffs_y, spline_lengths=None, incl_last_point=False, stepsize_approx=None,
                   stepnum_fixed=None):
    """
    Sample splines either with an approximately constant arc step (``stepsize_approx``; offline edges gen_edges.py:128,
    start spline OnlineTrajectoryHandler.py:248) or with a fixed number of uniform-t samples per segment
    (``stepnum_fixed``; online re-sample main_online_path_gen.py:313). Returns (path, spline_inds, t_values, dists).
    """
    if coeffs_x.shape != coeffs_y.shape:
        raise RuntimeError("Coefficient matrices must have the same length!")
    if coeffs_x.ndim == 2 and coeffs_x.shape[1] != 4:
        raise RuntimeError("Coefficient matrices do not have two dimensions!")
    if (stepsize_approx is None and stepnum_fixed is None) or (stepsize_approx is not None
                                                               and stepnum_fixed is not None):
        raise RuntimeError("Provide one of 'stepsize_approx' and 'stepnum_fixed' and set the other to 'None'!")
    if stepnum_fixed is not None and len(stepnum_fixed) != coeffs_x.shape[0]:
        raise RuntimeError("The provided list 'stepnum_fixed' must hold an entry for every spline!")

    if stepsize_approx is not None:
        if spline_lengths is None:
            spline_lengths = calc_spline_lengths(coeffs_x=coeffs_x, coeffs_y=coeffs_y, quickndirty=False)
        dists_cum = np.cumsum(spline_lengths)
        no_interp_points = math.ceil(dists_cum[-1] / stepsize_approx) + 1
        dists_interp = np.linspace(0.0, dists_cum[-1], no_interp_points)
    else:
        no_interp_points = sum(stepnum_fixed) - (len(stepnum_fixed) - 1)
        dists_interp = None

    path_interp = np.zeros((no_interp_points, 2))
    spline_inds = np.zeros(no_interp_points, dtype=int)
    t_values = np.zeros(no_interp_points)

    if stepsize_approx is not None:
        for i in range(no_interp_points - 1):
            j = np.argmax(dists_interp[i] < dists_cum)
            spline_inds[i] = j
            if j > 0:
                t_values[i] = (dists_interp[i] - dists_cum[j - 1]) / spline_lengths[j]
            else:
                if spline_lengths.ndim == 0:
                    t_values[i] = dists_interp[i] / spline_lengths
                else:
                    t_values[i] = dists_interp[i] / spline_lengths[0]
            path_interp[i, 0] = coeffs_x[j, 0] + coeffs_x[j, 1] * t_values[i] \
                + coeffs_x[j, 2] * math.pow(t_values[i], 2) + coeffs_x[j, 3] * math.pow(t_values[i], 3)
            path_interp[i, 1] = coeffs_y[j, 0] + coeffs_y[j, 1] * t_values[i] \
                + coeffs_y[j, 2] * math.pow(t_values[i], 2) + coeffs_y[j, 3] * math.pow(t_values[i], 3)
    else:
        j = 0
        for i in range(len(stepnum_fixed)):
            if i < len(stepnum_fixed) - 1:
                t_values[j:(j + stepnum_fixed[i] - 1)] = np.linspace(0, 1, stepnum_fixed[i])[:-1]
                spline_inds[j:(j + stepnum_fixed[i] - 1)] = i
                j += stepnum_fixed[i] - 1
            else:
                t_values[j:(j + stepnum_fixed[i])] = np.linspace(0, 1, stepnum_fixed[i])
                spline_inds[j:(j + stepnum_fixed[i])] = i
                j += stepnum_fixed[i]

        t_set = np.column_stack((np.ones(no_interp_points), t_values, np.power(t_values, 2), np.power(t_values, 3)))
        n_samples = np.array(stepnum_fixed)
        n_samples[:-1] -= 1
        path_interp[:, 0] = np.sum(np.multiply(np.repeat(coeffs_x, n_samples, axis=0), t_set), axis=1)
        path_interp[:, 1] = np.sum(np.multiply(np.repeat(coeffs_y, n_samples, axis=0), t_set), axis=1)

    if incl_last_point:
        path_interp[-1, 0] = np.sum(coeffs_x[-1])
        path_interp[-1, 1] = np.sum(coeffs_y[-1])
        spline_inds[-1] = coeffs_x.shape[0] - 1
        t_values[-1] = 1.0
    else:
        path_interp = path_interp[:-1]
        spline_inds = spline_inds[:-1]
        t_values = t_values[:-1]
        if dists_interp is not None:
            dists_interp = dists_interp[:-1]

    return path_interp, spline_inds, t_values, dists_interp
