"""
ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

NumPy restatement of the per-edge arithmetic of the offline build, in the reference's formulation:
  tph.calc_splines on two points (gen_edges.py:80-84; dense 4x4-per-axis system -> here the equivalent Hermite closed form is NOT
  used: the 8 x 8 linear system of the reference is solved), tph.calc_spline_lengths (15-point polyline),
  tph.interp_splines(stepsize_approx, incl_last_point=True), tph.calc_head_curv_an, the curvature filter (gen_edges.py:127-140),
  GraphBase.update_edge (GraphBase.py:421-436) and the curvature terms of gen_offline_cost.py:57-62.
Same call signature as ``offline_build.edges_on_device(...)`` so that the GPU-less container can run the HOST logic of
``build_lattice`` (skeleton, candidate edges, pruning, assembly) against the lattice exported from the reference's own GraphBase;
the HIP kernel is compared with this evaluator and with those lattices by the ``-m gpu`` tests.
"""
import math
import numpy as np


def evaluate(start, end, kappa_max_vel, raceline_edge, given_coeff, stepsize, kappa_max_turn, cap):
    n = start.shape[0]
    coeff = np.zeros((n, 8))
    # two-point spline with heading constraints: [a0, a1, a2, a3] from the 4 x 4 system of tph.calc_splines (one segment)
    M = np.array([[1.0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 0], [0, 1, 2, 3]])
    Minv = np.linalg.inv(M)
    el = np.sqrt(np.power(end[:, 0] - start[:, 0], 2) + np.power(end[:, 1] - start[:, 1], 2))
    for ax, fn in ((0, np.cos), (1, np.sin)):
        b = np.stack((start[:, ax], end[:, ax], fn(start[:, 2] + math.pi / 2) * el, fn(end[:, 2] + math.pi / 2) * el), axis=1)
        coeff[:, 4 * ax:4 * ax + 4] = b @ Minv.T
    rl = np.asarray(raceline_edge).astype(bool)
    coeff[rl] = np.asarray(given_coeff)[rl]
    cx, cy = coeff[:, 0:4], coeff[:, 4:8]

    def poly(c, t):
        return c[:, 0:1] + c[:, 1:2] * t + c[:, 2:3] * np.power(t, 2) + c[:, 3:4] * np.power(t, 3)
    t15 = np.linspace(0.0, 1.0, 15)[None, :]
    px, py = poly(cx, t15), poly(cy, t15)
    length15 = np.sum(np.sqrt(np.power(np.diff(px, axis=1), 2) + np.power(np.diff(py, axis=1), 2)), axis=1)
    ns = (np.ceil(length15 / stepsize) + 1).astype(np.int64)
    out = {"n_samples": ns.astype(np.int32), "valid": np.zeros(n, np.int32), "coeff": coeff, "length": np.zeros(n),
           "kappa_avg": np.zeros(n), "kappa_range": np.zeros(n), "samples": np.zeros((n, cap, 5))}
    for m in np.unique(ns):
        idx = np.flatnonzero(ns == m)
        if m > cap or m < 2:
            continue
        L = length15[idx][:, None]
        dists = np.linspace(0.0, 1.0, int(m))[None, :] * L            # np.linspace(0, len, m) row-wise
        dists = np.stack([np.linspace(0.0, float(v), int(m)) for v in length15[idx]])
        t = dists / L
        t[:, -1] = 1.0
        x, y = poly(cx[idx], t), poly(cy[idx], t)
        x[:, -1], y[:, -1] = np.sum(cx[idx], axis=1), np.sum(cy[idx], axis=1)
        xd = cx[idx][:, 1:2] + 2 * cx[idx][:, 2:3] * t + 3 * cx[idx][:, 3:4] * np.power(t, 2)
        yd = cy[idx][:, 1:2] + 2 * cy[idx][:, 2:3] * t + 3 * cy[idx][:, 3:4] * np.power(t, 2)
        xdd = 2 * cx[idx][:, 2:3] + 6 * cx[idx][:, 3:4] * t
        ydd = 2 * cy[idx][:, 2:3] + 6 * cy[idx][:, 3:4] * t
        psi = np.arctan2(yd, xd) - math.pi / 2
        psi = np.sign(psi) * np.mod(np.abs(psi), 2 * math.pi)
        psi = np.where(psi >= math.pi, psi - 2 * math.pi, psi)
        psi = np.where(psi < -math.pi, psi + 2 * math.pi, psi)
        kappa = (xd * ydd - yd * xdd) / np.power(np.power(xd, 2) + np.power(yd, 2), 1.5)
        elen = np.sqrt(np.power(np.diff(x, axis=1), 2) + np.power(np.diff(y, axis=1), 2))
        out["samples"][idx, :m, 0], out["samples"][idx, :m, 1] = x, y
        out["samples"][idx, :m, 2], out["samples"][idx, :m, 3] = psi, kappa
        out["samples"][idx, :m - 1, 4] = elen
        ok = np.all(np.abs(kappa) <= kappa_max_turn, axis=1) & np.all(np.abs(kappa) <= np.asarray(kappa_max_vel)[idx][:, None], axis=1)
        out["valid"][idx] = (ok | rl[idx]).astype(np.int32)
        out["length"][idx] = np.sum(elen, axis=1)
        out["kappa_avg"][idx] = np.sum(np.abs(kappa), axis=1) / float(m)
        out["kappa_range"][idx] = np.abs(np.max(kappa, axis=1) - np.min(kappa, axis=1))
    return out
