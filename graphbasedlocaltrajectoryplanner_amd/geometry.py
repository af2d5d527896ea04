"""
Small host-side geometry helpers of the hot path (row U1 of SURVEY.md section 8a): projection of a point on a polyline.
Semantics follow graph_ltpl/helper_funcs/src/get_s_coord.py:8-121 and closest_path_index.py:4-32.
"""
import math
import numpy as np


def angle3pt(a, b, c) -> float:
    """Signed angle of the turn a -> c around b, wrapped to (-pi, pi] (get_s_coord.py:102-121)."""
    ang = math.atan2(c[1] - b[1], c[0] - b[0]) - math.atan2(a[1] - b[1], a[0] - b[0])
    if ang > math.pi:
        ang -= 2 * math.pi
    elif ang <= -math.pi:
        ang += 2 * math.pi
    return ang


def closest_path_index(path: np.ndarray, pos) -> int:
    """Index of the path point closest to ``pos`` (closest_path_index.py:29-30 with n_closest=1)."""
    d2 = np.power(path[:, 0] - pos[0], 2) + np.power(path[:, 1] - pos[1], 2)
    return int(np.argmin(d2))


def get_s_coord(ref_line: np.ndarray, pos, s_array: np.ndarray = None, only_index: bool = False,
                closed: bool = False):
    """
    s coordinate of ``pos`` along ``ref_line`` and the two neighbouring indices (get_s_coord.py:8-99): closest point,
    the neighbour on the side with the larger |angle3pt|, perpendicular foot on that segment.
    """
    n = ref_line.shape[0]
    nb = closest_path_index(ref_line, pos)
    if closed:
        i1 = nb - 1                       # may be -1: Python indexing wraps to the last point (get_s_coord.py:39)
        i2 = nb + 1
        if i2 > n - 1:
            i2 = 0
    else:
        i1 = max(nb - 1, 0)
        i2 = min(nb + 1, n - 1)

    ang1 = abs(angle3pt(ref_line[nb, :], pos, ref_line[i1, :]))
    ang2 = abs(angle3pt(ref_line[nb, :], pos, ref_line[i2, :]))

    s = None
    if not only_index:
        if ang1 > ang2:
            a_pos, b_pos = ref_line[i1, :], ref_line[nb, :]
        else:
            a_pos, b_pos = ref_line[nb, :], ref_line[i2, :]
        if s_array is None:
            s_array = np.cumsum(np.sqrt(np.sum(np.power(np.diff(ref_line, axis=0), 2), axis=1)))
        if s_array[0] > 0.05:
            s_array = np.insert(s_array, 0, 0.0)
        t = ((pos[0] - a_pos[0]) * (b_pos[0] - a_pos[0]) + (pos[1] - a_pos[1]) * (b_pos[1] - a_pos[1])) / \
            (np.power(b_pos[0] - a_pos[0], 2) + np.power(b_pos[1] - a_pos[1], 2))
        foot = [a_pos[0] + t * (b_pos[0] - a_pos[0]), a_pos[1] + t * (b_pos[1] - a_pos[1])]
        ds = np.sqrt(np.power(a_pos[0] - foot[0], 2) + np.power(a_pos[1] - foot[1], 2))
        s = (s_array[i1] if ang1 > ang2 else s_array[nb]) + ds

    idx = [i1, nb] if ang1 >= ang2 else [nb, i2]
    return s, idx
