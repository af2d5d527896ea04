import math
import numpy as np
from .normalize_psi import normalize_psi


def calc_head_curv_an(coeffs_x, coeffs_y, ind_spls, t_spls, calc_curv=True, calc_dcurv=False):
    """Analytic heading (0 = north) and curvature of cubic splines at (spline index, t) pairs (tph)."""
    if coeffs_x.shape[0] != coeffs_y.shape[0]:
        raise ValueError("Coefficient matrices must have the same length!")
    if ind_spls.size != t_spls.size:
        raise ValueError("ind_spls and t_spls must have the same length!")

    x_d = coeffs_x[ind_spls, 1] + 2 * coeffs_x[ind_spls, 2] * t_spls + 3 * coeffs_x[ind_spls, 3] * np.power(t_spls, 2)
    y_d = coeffs_y[ind_spls, 1] + 2 * coeffs_y[ind_spls, 2] * t_spls + 3 * coeffs_y[ind_spls, 3] * np.power(t_spls, 2)

    psi = np.arctan2(y_d, x_d) - math.pi / 2
    psi = normalize_psi(psi)

    if calc_curv:
        x_dd = 2 * coeffs_x[ind_spls, 2] + 6 * coeffs_x[ind_spls, 3] * t_spls
        y_dd = 2 * coeffs_y[ind_spls, 2] + 6 * coeffs_y[ind_spls, 3] * t_spls
        kappa = (x_d * y_dd - y_d * x_dd) / np.power(np.power(x_d, 2) + np.power(y_d, 2), 1.5)
    else:
        kappa = 0.0

    if calc_dcurv:
        raise NotImplementedError("calc_dcurv is not used by the reference and not restated in the oracle shim")

    return psi, kappa
