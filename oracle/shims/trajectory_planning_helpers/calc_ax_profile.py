import numpy as np


def calc_ax_profile(vx_profile, el_lengths, eq_length_output=False):
    """ax_i = (v_{i+1}^2 - v_i^2) / (2 * el_i) (tph calc_ax_profile); optional trailing zero for equal length."""
    if vx_profile.size != el_lengths.size + 1:
        raise RuntimeError("Array size of vx_profile should be 1 element bigger than el_lengths!")
    if eq_length_output:
        ax_profile = np.zeros(vx_profile.size)
        ax_profile[:-1] = (np.power(vx_profile[1:], 2) - np.power(vx_profile[:-1], 2)) / (2 * el_lengths)
    else:
        ax_profile = (np.power(vx_profile[1:], 2) - np.power(vx_profile[:-1], 2)) / (2 * el_lengths)
    return ax_profile
