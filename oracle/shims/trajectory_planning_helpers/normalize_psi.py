import math
import numpy as np


def normalize_psi(psi):
    """Map heading(s) to [-pi, pi): remove multiples of 2*pi keeping the sign, then wrap (tph normalize_psi)."""
    psi_out = np.sign(psi) * np.mod(np.abs(psi), 2 * math.pi)
    if type(psi_out) is np.ndarray:
        psi_out[psi_out >= math.pi] -= 2 * math.pi
        psi_out[psi_out < -math.pi] += 2 * math.pi
    else:
        if psi_out >= math.pi:
            psi_out -= 2 * math.pi
        elif psi_out < -math.pi:
            psi_out += 2 * math.pi
    return psi_out
