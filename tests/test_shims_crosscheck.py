"""
Independent cross-checks of the restated third-party arithmetic (oracle/shims) -- the part of the parity chain that the
reference's own tests do not pin (SURVEY.md sections 4 and 8c):

  * tph.calc_splines (dense 4N x 4N formulation)  vs  scipy.interpolate.CubicSpline with clamped end slopes in the
    cumulated el_lengths parameter (identity of SURVEY.md App. B.1), and vs the two-point Hermite closed form
  * igraph shim Dijkstra                            vs  scipy.sparse.csgraph.dijkstra on the Monteblanco lattice graph
    (path COST must agree; node lists may only differ on exact ties, which are counted)
  * tph.calc_vel_profile / calc_vel_profile_brake   vs  closed-form cases (constant curvature, constant deceleration)
  * tph.interp_splines / calc_head_curv_an          vs  a circle (heading, curvature, sample counts)
"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(ROOT, "oracle", "shims")
if SHIMS not in sys.path:
    sys.path.insert(0, SHIMS)

import trajectory_planning_helpers as tph          # noqa: E402  (the shim)
import igraph                                      # noqa: E402  (the shim)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_calc_splines_equals_clamped_cubic_spline(seed):
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 30))
    ang = np.cumsum(rng.uniform(-0.25, 0.25, n))
    step = rng.uniform(5.0, 35.0, n)
    path = np.vstack((np.zeros(2), np.cumsum(np.column_stack((np.cos(ang) * step, np.sin(ang) * step)), axis=0)))
    el = np.hypot(*np.diff(path, axis=0).T) * rng.uniform(1.0, 1.08, n)            # spline lengths >= chord lengths
    psi_s, psi_e = ang[0] - np.pi / 2 + rng.uniform(-0.1, 0.1), ang[-1] - np.pi / 2 + rng.uniform(-0.1, 0.1)
    cx, cy, _, _ = tph.calc_splines.calc_splines(path=path, el_lengths=el, psi_s=psi_s, psi_e=psi_e)
    u = np.concatenate(([0.0], np.cumsum(el)))
    for col, coeffs, f in ((0, cx, np.cos), (1, cy, np.sin)):
        cs = CubicSpline(u, path[:, col], bc_type=((1, f(psi_s + np.pi / 2)), (1, f(psi_e + np.pi / 2))))
        # scipy: c[k, i] multiplies (x - u_i)^(3 - k); tph parameter t = (x - u_i) / el_i
        exp = np.column_stack((cs.c[3], cs.c[2] * el, cs.c[1] * el ** 2, cs.c[0] * el ** 3))
        assert np.max(np.abs(coeffs - exp)) <= 1e-9 * max(1.0, np.max(np.abs(exp)))


def test_two_point_spline_is_cubic_hermite():
    p0, p1 = np.array([3.0, -2.0]), np.array([11.0, 4.5])
    psi0, psi1 = 0.3, -0.2
    d = float(np.hypot(*(p1 - p0)))
    cx, cy, _, _ = tph.calc_splines.calc_splines(path=np.vstack((p0, p1)), psi_s=psi0, psi_e=psi1)
    t0 = d * np.array([np.cos(psi0 + np.pi / 2), np.sin(psi0 + np.pi / 2)])
    t1 = d * np.array([np.cos(psi1 + np.pi / 2), np.sin(psi1 + np.pi / 2)])
    dl = p1 - p0
    exp = np.array([p0, t0, 3 * dl - 2 * t0 - t1, -2 * dl + t0 + t1])               # a0 .. a3 (x, y)
    assert np.allclose(cx[0], exp[:, 0], rtol=1e-12, atol=1e-12) and np.allclose(cy[0], exp[:, 1], rtol=1e-12, atol=1e-12)


def test_interp_and_head_curv_on_a_circle():
    R, n = 50.0, 24
    a = np.linspace(0.0, np.pi / 2, n + 1)
    path = np.column_stack((R * np.cos(a), R * np.sin(a)))
    psi_s, psi_e = a[0], a[-1]                                                     # tangent angle = a + pi/2, psi = tangent - pi/2
    cx, cy, _, _ = tph.calc_splines.calc_splines(path=path, psi_s=psi_s, psi_e=psi_e)
    steps = np.full(n, 5)
    pts, inds, tvals, _ = tph.interp_splines.interp_splines(coeffs_x=cx, coeffs_y=cy, incl_last_point=True, stepnum_fixed=list(steps))
    assert pts.shape[0] == int(np.sum(steps) - (n - 1))                            # shared knots are not repeated
    assert np.max(np.abs(np.hypot(pts[:, 0], pts[:, 1]) - R)) < 1e-3
    psi, kappa = tph.calc_head_curv_an.calc_head_curv_an(coeffs_x=cx, coeffs_y=cy, ind_spls=inds, t_spls=tvals)
    assert np.max(np.abs(kappa - 1.0 / R)) < 1e-4
    ang = np.arctan2(pts[:, 1], pts[:, 0])
    assert np.max(np.abs(np.mod(psi - ang + np.pi, 2 * np.pi) - np.pi)) < 1e-3


def test_igraph_shim_dijkstra_cost_equals_scipy(monteblanco):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    lat = monteblanco
    sl, sn, dlr, dn = lat.edge_endpoints()
    src = lat.layer_off[sl] + sn
    dst = lat.layer_off[dlr] + dn
    V = lat.num_nodes
    g = igraph.Graph()
    g.to_directed()
    for v in range(V):
        g.add_vertex(name=str(v))
    for e in range(lat.num_edges):
        g.add_edge(source=str(int(src[e])), target=str(int(dst[e])), offline_cost=float(lat.edge_cost[e]))
    m = csr_matrix((lat.edge_cost + 1e-300, (src, dst)), shape=(V, V))             # keep explicit zero-cost edges
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(12):
        l0 = int(rng.integers(0, lat.num_layers))
        v0 = int(lat.layer_off[l0] + rng.integers(0, lat.nodes_in_layer[l0]))
        l1 = (l0 + int(rng.integers(5, 25))) % lat.num_layers
        v1 = int(lat.layer_off[l1] + rng.integers(0, lat.nodes_in_layer[l1]))
        d = dijkstra(m, directed=True, indices=v0)[v1]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vpath = g.get_shortest_paths(str(v0), to=str(v1), weights="offline_cost", output="vpath")[0]
        if not np.isfinite(d):
            assert vpath == []
            continue
        cost = 0.0
        for a, b in zip(vpath[:-1], vpath[1:]):
            cost += float(lat.edge_cost[lat.find_edge(int(lat.layer_of(a)), int(a - lat.layer_off[lat.layer_of(a)]),
                                                      int(lat.layer_of(b)), int(b - lat.layer_off[lat.layer_of(b)]))])
        assert vpath[0] == v0 and vpath[-1] == v1
        assert abs(cost - d) <= 1e-9 * max(1.0, d)
        checked += 1
    assert checked >= 6


def test_vel_profile_closed_forms():
    n = 200
    el = np.full(n - 1, 2.0)
    axm = np.array([[0.0, 50.0], [200.0, 50.0]])
    # (a) constant curvature, generous longitudinal limits, start and end at the lateral limit: v = sqrt(ay / kappa)
    kappa = np.full(n, 0.02)
    v_lat = np.sqrt(8.0 / 0.02)
    vx = tph.calc_vel_profile.calc_vel_profile(ax_max_machines=axm, kappa=kappa, el_lengths=el, closed=False, drag_coeff=0.0,
                                               m_veh=1000.0, loc_gg=np.column_stack((np.full(n, 6.0), np.full(n, 8.0))),
                                               v_max=100.0, dyn_model_exp=1.0, v_start=v_lat, v_end=v_lat)
    assert np.allclose(vx, v_lat, rtol=1e-12)
    # (b) straight line, no drag: constant acceleration a from v_start, then constant deceleration to v_end
    kappa0 = np.zeros(n)
    a, v0, v1 = 4.0, 10.0, 5.0
    vx = tph.calc_vel_profile.calc_vel_profile(ax_max_machines=np.array([[0.0, a], [200.0, a]]), kappa=kappa0, el_lengths=el,
                                               closed=False, drag_coeff=0.0, m_veh=1000.0,
                                               loc_gg=np.column_stack((np.full(n, a), np.full(n, 8.0))), v_max=1000.0,
                                               dyn_model_exp=1.0, v_start=v0, v_end=v1)
    s = np.concatenate(([0.0], np.cumsum(el)))
    v_acc = np.sqrt(v0 ** 2 + 2 * a * s)
    v_dec = np.sqrt(v1 ** 2 + 2 * a * (s[-1] - s))
    assert np.allclose(vx, np.minimum(v_acc, v_dec), rtol=1e-9)
    # (c) brake profile on a straight: v^2 = v0^2 - 2 a s until standstill, zeros behind
    vb = tph.calc_vel_profile_brake.calc_vel_profile_brake(kappa=kappa0, el_lengths=el, v_start=30.0, drag_coeff=0.0, m_veh=1000.0,
                                                           loc_gg=np.column_stack((np.full(n, a), np.full(n, 8.0))), dyn_model_exp=1.0)
    rad = 30.0 ** 2 - 2 * a * s
    k = int(np.argmax(rad < 0.0))
    assert np.allclose(vb[:k], np.sqrt(rad[:k]), rtol=1e-9) and np.all(vb[k:] == 0.0)
