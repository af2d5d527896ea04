import math
import numpy as np


def calc_spline_lengths(coeffs_x, coeffs_y, quickndirty=False, no_interp_points=15):
    """Per-spline length: chord (quickndirty) or polyline over ``no_interp_points`` uniform-t samples (tph)."""
    if coeffs_x.size == 4 and coeffs_x.shape[0] == 4:
        coeffs_x = np.expand_dims(coeffs_x, 0)
        coeffs_y = np.expand_dims(coeffs_y, 0)

    no_splines = coeffs_x.shape[0]
    spline_lengths = np.zeros(no_splines)

    if quickndirty:
        for i in range(no_splines):
            spline_lengths[i] = math.sqrt(math.pow(np.sum(coeffs_x[i]) - coeffs_x[i, 0], 2)
                                          + math.pow(np.sum(coeffs_y[i]) - coeffs_y[i, 0], 2))
    else:
        t_steps = np.linspace(0.0, 1.0, no_interp_points)
        spl_coords = np.zeros((no_interp_points, 2))
        for i in range(no_splines):
            spl_coords[:, 0] = coeffs_x[i, 0] + coeffs_x[i, 1] * t_steps + coeffs_x[i, 2] * np.power(t_steps, 2) \
                + coeffs_x[i, 3] * np.power(t_steps, 3)
            spl_coords[:, 1] = coeffs_y[i, 0] + coeffs_y[i, 1] * t_steps + coeffs_y[i, 2] * np.power(t_steps, 2) \
                + coeffs_y[i, 3] * np.power(t_steps, 3)
            spline_lengths[i] = np.sum(np.sqrt(np.sum(np.power(np.diff(spl_coords, axis=0), 2), axis=1)))

    return spline_lengths
