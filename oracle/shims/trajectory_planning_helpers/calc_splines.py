import math
import numpy as np


def calc_splines(path, el_lengths=None, psi_s=None, psi_e=None, use_dist_scaling=True):
    """
    Curvature-continuous cubic splines x_i(t), y_i(t), t in [0, 1], through ``path`` (tph calc_splines).

    Unknowns per segment [a0, a1, a2, a3]; rows per segment: position at t=0, position at t=1, heading continuity
    (scaled by el_i / el_{i+1}), curvature continuity (scaled by its square). Unclosed paths fix the start / end
    heading (psi = 0 is north, hence the +pi/2) scaled by the first / last element length; closed paths (first == last
    point and no psi_s) get periodic heading / curvature rows. Dense 4N x 4N solve, once for x and once for y.
    Call sites in the reference: main_online_path_gen.py:305, OnlineTrajectoryHandler.py:244, gen_edges.py:47,88.
    """
    closed = bool(np.all(np.isclose(path[0], path[-1])) and psi_s is None)

    if not closed and (psi_s is None or psi_e is None):
        raise RuntimeError("Headings must be provided for unclosed spline calculation!")
    if el_lengths is not None and path.shape[0] != el_lengths.size + 1:
        raise RuntimeError("el_lengths input must be one element smaller than path input!")

    if use_dist_scaling and el_lengths is None:
        el_lengths = np.sqrt(np.sum(np.power(np.diff(path, axis=0), 2), axis=1))
    elif el_lengths is not None:
        el_lengths = np.copy(el_lengths)

    if use_dist_scaling and closed:
        el_lengths = np.append(el_lengths, el_lengths[0])

    no_splines = path.shape[0] - 1

    if use_dist_scaling:
        scaling = el_lengths[:-1] / el_lengths[1:]
    else:
        scaling = np.ones(no_splines - 1)

    M = np.zeros((no_splines * 4, no_splines * 4))
    b_x = np.zeros((no_splines * 4, 1))
    b_y = np.zeros((no_splines * 4, 1))

    block = np.array([[1, 0, 0, 0, 0, 0, 0, 0],
                      [1, 1, 1, 1, 0, 0, 0, 0],
                      [0, 1, 2, 3, 0, -1, 0, 0],
                      [0, 0, 2, 6, 0, 0, -2, 0]], dtype=float)

    for i in range(no_splines):
        j = i * 4
        if i < no_splines - 1:
            M[j: j + 4, j: j + 8] = block
            M[j + 2, j + 5] *= scaling[i]
            M[j + 3, j + 6] *= math.pow(scaling[i], 2)
        else:
            M[j: j + 2, j: j + 4] = [[1, 0, 0, 0],
                                     [1, 1, 1, 1]]
        b_x[j: j + 2] = [[path[i, 0]], [path[i + 1, 0]]]
        b_y[j: j + 2] = [[path[i, 1]], [path[i + 1, 1]]]

    if not closed:
        M[-2, 1] = 1
        el_length_s = 1.0 if el_lengths is None else el_lengths[0]
        b_x[-2] = math.cos(psi_s + math.pi / 2) * el_length_s
        b_y[-2] = math.sin(psi_s + math.pi / 2) * el_length_s

        M[-1, -4:] = [0, 1, 2, 3]
        el_length_e = 1.0 if el_lengths is None else el_lengths[-1]
        b_x[-1] = math.cos(psi_e + math.pi / 2) * el_length_e
        b_y[-1] = math.sin(psi_e + math.pi / 2) * el_length_e
    else:
        M[-2, 1] = scaling[-1]
        M[-2, -3:] = [-1, -2, -3]
        M[-1, 2] = 2 * math.pow(scaling[-1], 2)
        M[-1, -2:] = [-2, -6]

    x_les = np.squeeze(np.linalg.solve(M, b_x))
    y_les = np.squeeze(np.linalg.solve(M, b_y))

    coeffs_x = np.reshape(x_les, (no_splines, 4))
    coeffs_y = np.reshape(y_les, (no_splines, 4))

    normvec = np.stack((coeffs_y[:, 1], -coeffs_x[:, 1]), axis=1)
    norm_factors = 1.0 / np.sqrt(np.sum(np.power(normvec, 2), axis=1))
    normvec_normalized = np.expand_dims(norm_factors, axis=1) * normvec

    return coeffs_x, coeffs_y, M, normvec_normalized
