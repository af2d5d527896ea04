"""
Offline lattice build: from a global race line file + the offline parameters straight to the
struct-of-arrays ``Lattice`` that ``ltpl_create`` uploads -- no igraph object, no pickle (``Lattice.save`` / ``Lattice.load`` is the
portable on-disk format).

What the reference does in graph_ltpl/offline_graph/src/main_offline_callback.py:57-196:
  import_globtraj_csv -> variable_step_size -> gen_node_skeleton -> gen_edges (splines, sampling, curvature filter)
  -> prune_graph -> gen_offline_cost
Here the O(#layers) parts (layer selection, node skeleton, closed race line spline) are vectorised NumPy on the host, and the
per-edge arithmetic -- the reference's hot spot: ~36 k two-point splines sampled one by one in Python, 21 s of the 24 s build --
runs for ALL candidate edges in one launch of ``k_offline_edges`` (C ABI ``ltpl_offline_edges``, lane = edge). Pruning of dead ends
is a fixed-point iteration on in/out-degree arrays.

    track  = import_track_csv(path)                      # or any dict with the same arrays
    lat    = build_lattice(track, OFFLINE_DEFAULTS, edges_on_device(lib))
"""
import ctypes as C
import math
import numpy as np

from . import _capi
from ._capi import _pf64, _pi32, _p, _f64, _i32
from .lattice import Lattice

# params/ltpl_config_offline.ini of the reference
OFFLINE_DEFAULTS = dict(lat_resolution=0.5, variable_heading=True, lon_straight_step=30.0, lon_curve_step=10.0, curve_thr=0.008,
                        lat_offset=0.25, virt_goal_n=True, min_vel_race=0.5, closure_detection_dist=20.0, vel_decrease_lat=0.1,
                        min_plan_horizon=300.0, plan_horizon_mode="distance", stepsize_approx=2.5, veh_width=2.8, veh_length=4.7,
                        veh_turn=7.0, w_raceline=1.0, w_raceline_sat=1.0, w_length=0.0, w_curv_avg=7500.0, w_curv_peak=2500.0,
                        w_virt_goal=10000.0)


def config_from_ini(path) -> dict:
    """The keys main_offline_callback.py / gen_offline_cost.py read from params/ltpl_config_offline.ini."""
    import configparser
    ini = configparser.ConfigParser()
    if not ini.read(path):
        raise ValueError('Specified graph config file does not exist or is empty!')
    c = dict(OFFLINE_DEFAULTS)
    for sec, keys in (("LATTICE", ("lat_resolution", "lon_straight_step", "lon_curve_step", "curve_thr", "lat_offset", "min_vel_race",
                                   "closure_detection_dist")),
                      ("PLANNINGTARGET", ("vel_decrease_lat", "min_plan_horizon")), ("SAMPLING", ("stepsize_approx",)),
                      ("VEHICLE", ("veh_width", "veh_length", "veh_turn")),
                      ("COST", ("w_raceline", "w_raceline_sat", "w_length", "w_curv_avg", "w_curv_peak", "w_virt_goal"))):
        for k in keys:
            c[k] = ini.getfloat(sec, k)
    c["variable_heading"] = ini.getboolean("LATTICE", "variable_heading")
    c["virt_goal_n"] = ini.getboolean("LATTICE", "virt_goal_n")
    c["plan_horizon_mode"] = ini.get("PLANNINGTARGET", "plan_horizon_mode")
    return c


def import_track_csv(path) -> dict:
    """Columns of the global race line file (imp_global_traj/src/import_globtraj_csv.py:4-36): the last row only closes the s axis."""
    d = np.loadtxt(path, delimiter=';')
    return {"refline": d[:-1, 0:2], "width_right": d[:-1, 2], "width_left": d[:-1, 3], "normvec": d[:-1, 4:6], "alpha": d[:-1, 6],
            "length_rl": np.diff(d[:, 7]), "kappa_rl": d[:-1, 9], "vel_rl": d[:-1, 10]}


def _wrap_angle(a):
    a = np.asarray(a, dtype=float)
    out = np.sign(a) * np.mod(np.abs(a), 2 * math.pi)
    out = np.where(out >= math.pi, out - 2 * math.pi, out)
    return np.where(out < -math.pi, out + 2 * math.pi, out)


def _layer_rows(kappa, dist, d_curve, d_straight, curve_th, force_last):
    """Rows of the race line file that host a layer (variable_step_size.py:4-63): tighter spacing in curves."""
    rows, nxt, nxt_min, cur = [], 0.0, 0.0, 0.0
    for i in range(len(dist)):
        ahead = cur + dist[i]
        curve = abs(kappa[i]) > curve_th
        if ahead > nxt_min and curve:
            nxt = cur
        if ahead > nxt:
            rows.append(i)
            nxt += d_straight if abs(kappa[i]) < curve_th else d_curve
            nxt_min = cur + d_curve
        cur = ahead
    if force_last and len(kappa) - 1 not in rows:
        rows.append(len(kappa) - 1)
    return np.array(rows, dtype=np.int64)


def _closed_line_heading(pts, el):
    """Heading (0 = north) of a closed polyline by central differences over +-k points, k from a 1 m preview / review window
    (tph.calc_head_curv_num as gen_node_skeleton.py:63-90 calls it -- always as a CLOSED line, also on open tracks)."""
    k = max(int(round(1.0 / float(np.average(el)))), 1)
    ahead, behind = np.roll(pts, -k, axis=0), np.roll(pts, k, axis=0)
    return _wrap_angle(np.arctan2(ahead[:, 1] - behind[:, 1], ahead[:, 0] - behind[:, 0]) - math.pi / 2)


def _periodic_spline_coeffs(pts):
    """Coefficients per segment (t in [0, 1]) of the closed C2 spline through ``pts`` + the first point again, chord-length scaled
    (tph.calc_splines(path=raceline_cl), gen_edges.py:40-42): a cyclic tridiagonal system in the knot slopes."""
    n = pts.shape[0]
    nxt = np.roll(pts, -1, axis=0)
    h = np.hypot(nxt[:, 0] - pts[:, 0], nxt[:, 1] - pts[:, 1])
    ih, ih_prev = 1.0 / h, 1.0 / np.roll(h, 1)
    A = np.zeros((n, n))
    idx = np.arange(n)
    A[idx, idx] = 2.0 * (ih_prev + ih)
    A[idx, (idx - 1) % n] += ih_prev
    A[idx, (idx + 1) % n] += ih
    delta = nxt - pts
    rhs = 3.0 * (np.roll(delta, 1, axis=0) * (ih_prev ** 2)[:, None] + delta * (ih ** 2)[:, None])
    m = np.linalg.solve(A, rhs)
    t0, t1 = m * h[:, None], np.roll(m, -1, axis=0) * h[:, None]
    a2, a3 = 3.0 * delta - 2.0 * t0 - t1, -2.0 * delta + t0 + t1
    return np.column_stack((pts[:, 0], t0[:, 0], a2[:, 0], a3[:, 0], pts[:, 1], t0[:, 1], a2[:, 1], a3[:, 1]))


def edges_on_device(lib, device=-1):
    """Edge evaluator bound to ``ltpl_offline_edges`` of libltpl_hip.so (``lib`` = ctypes handle, e.g. ``HipBackend.lib``)."""
    class EdgesIn(C.Structure):
        _fields_ = [("n_edges", C.c_int32), ("cap_samples", C.c_int32), ("stepsize_approx", C.c_double),
                    ("kappa_max_turn", C.c_double), ("start_x", _pf64), ("start_y", _pf64), ("start_psi", _pf64),
                    ("end_x", _pf64), ("end_y", _pf64), ("end_psi", _pf64), ("kappa_max_vel", _pf64),
                    ("raceline_edge", _pi32), ("given_coeff", _pf64)]

    class EdgesOut(C.Structure):
        _fields_ = [("n_samples", _pi32), ("valid", _pi32), ("coeff", _pf64), ("length", _pf64), ("kappa_avg", _pf64),
                    ("kappa_range", _pf64), ("samples", _pf64)]
    lib.ltpl_offline_edges.argtypes = [C.c_int, C.POINTER(EdgesIn), C.POINTER(EdgesOut)]
    lib.ltpl_last_error.argtypes = [C.c_void_p]
    lib.ltpl_last_error.restype = C.c_char_p

    def evaluate(start, end, kappa_max_vel, raceline_edge, given_coeff, stepsize, kappa_max_turn, cap):
        n = start.shape[0]
        keep = [_f64(start[:, 0]), _f64(start[:, 1]), _f64(start[:, 2]), _f64(end[:, 0]), _f64(end[:, 1]), _f64(end[:, 2]),
                _f64(kappa_max_vel), _i32(raceline_edge), _f64(given_coeff)]
        o = {"n_samples": np.zeros(n, np.int32), "valid": np.zeros(n, np.int32), "coeff": np.zeros((n, 8)), "length": np.zeros(n),
             "kappa_avg": np.zeros(n), "kappa_range": np.zeros(n), "samples": np.zeros((n, cap, 5))}
        i, out = EdgesIn(), EdgesOut()
        i.n_edges, i.cap_samples, i.stepsize_approx, i.kappa_max_turn = n, int(cap), float(stepsize), float(kappa_max_turn)
        (i.start_x, i.start_y, i.start_psi, i.end_x, i.end_y, i.end_psi, i.kappa_max_vel) = (_p(a, _pf64) for a in keep[:7])
        i.raceline_edge, i.given_coeff = _p(keep[7], _pi32), _p(keep[8], _pf64)
        out.n_samples, out.valid = _p(o["n_samples"], _pi32), _p(o["valid"], _pi32)
        for k in ("coeff", "length", "kappa_avg", "kappa_range", "samples"):
            setattr(out, k, _p(o[k], _pf64))
        rc = lib.ltpl_offline_edges(int(device), C.byref(i), C.byref(out))
        if rc != 0:
            raise _capi.BackendError("ltpl_offline_edges failed (%s): %s" % (_capi._STATUS.get(rc, rc),
                                                                          (lib.ltpl_last_error(None) or b"").decode()))
        return o
    return evaluate


def build_lattice(track: dict, cfg: dict, evaluate_edges) -> Lattice:
    """``track``: arrays of ``import_track_csv``; ``cfg``: offline parameters; ``evaluate_edges``: the per-edge arithmetic
    (``edges_on_device`` in the product)."""
    if cfg["lat_offset"] <= 0:
        raise ValueError('Requested to small lateral offset! A lateral offset larger than zero must be allowed!')
    refline, normvec, alpha = np.asarray(track["refline"], float), np.asarray(track["normvec"], float), np.asarray(track["alpha"], float)
    w_r, w_l = np.asarray(track["width_right"], float), np.asarray(track["width_left"], float)
    length_rl, kappa_rl, vel_rl = (np.asarray(track[k], float) for k in ("length_rl", "kappa_rl", "vel_rl"))

    # ---- global race line + layer selection (main_offline_callback.py:84-125) ------------------------------------------------
    s_full = np.concatenate(([0.0], np.cumsum(length_rl)))
    xy = refline + normvec * alpha[:, None]
    rl_params = np.column_stack((xy, kappa_rl, vel_rl))
    closed = bool(np.hypot(xy[0, 0] - xy[-1, 0], xy[0, 1] - xy[-1, 1]) < cfg["closure_detection_dist"])
    glob_rl = (np.column_stack((s_full, np.vstack((rl_params, rl_params[0])))) if closed
               else np.column_stack((s_full[:-1], rl_params)))
    rows = _layer_rows(kappa_rl, length_rl, cfg["lon_curve_step"], cfg["lon_straight_step"], cfg["curve_thr"], not closed)
    refline, normvec, alpha, w_r, w_l, vel_l = refline[rows], normvec[rows], alpha[rows], w_r[rows], w_l[rows], vel_rl[rows]
    s_layers = s_full[rows]
    L = rows.size
    seg_len = np.array([np.sum(length_rl[a:b]) for a, b in zip(rows[:-1], rows[1:])] + [0.0])

    # ---- node skeleton (gen_node_skeleton.py:38-166): lateral grid per layer, heading blended towards the bounds -----------------
    vw2, res = cfg["veh_width"] / 2, cfg["lat_resolution"]
    margin = min(float(np.min(w_l - vw2 + alpha)), float(np.min(w_r - vw2 - alpha)))
    if margin < 0.0:
        raise ValueError("Provided raceline holds points outside the safety margin! Reduce the vehicle width or adapt the race "
                         "line (maximum possible vehicle width: %.3fm)." % (cfg["veh_width"] + 2 * margin))
    raceline = refline + normvec * alpha[:, None]
    psi_rl = _closed_line_heading(raceline, seg_len)
    if cfg["variable_heading"]:
        b_r, b_l = refline + normvec * w_r[:, None], refline - normvec * w_l[:, None]
        psi_bl = _closed_line_heading(b_l, np.hypot(*(np.roll(b_l, -1, axis=0) - b_l).T))
        psi_br = _closed_line_heading(b_r, np.hypot(*(np.roll(b_r, -1, axis=0) - b_r).T))
    rl_idx = np.floor((w_l - vw2 + alpha) / res).astype(np.int64)
    node_pos, node_psi, nodes_in_layer = [], [], np.zeros(L, np.int64)

    def blend(a, b, num, drop_last):
        if abs(a - b) < math.pi:
            v = np.linspace(a, b, num=num)
        else:
            v = _wrap_angle(np.linspace(a + 2 * math.pi * (a < 0), b + 2 * math.pi * (b < 0), num=num))
        return v[:-1] if drop_last else v
    for i in range(L):
        lat_coords = np.arange(alpha[i] - rl_idx[i] * res, w_r[i] - vw2, res)
        nodes_in_layer[i] = lat_coords.size
        node_pos.append(refline[i][None, :] + normvec[i][None, :] * lat_coords[:, None])
        if cfg["variable_heading"]:
            node_psi.append(np.append(blend(psi_bl[i], psi_rl[i], int(rl_idx[i]) + 1, True),
                                      blend(psi_rl[i], psi_br[i], lat_coords.size - int(rl_idx[i]), False)))
        else:
            node_psi.append(np.repeat(psi_rl[i], lat_coords.size))
    layer_off = np.concatenate(([0], np.cumsum(nodes_in_layer)))
    node_pos, node_psi = np.vstack(node_pos), np.concatenate(node_psi)
    V = int(layer_off[-1])

    # ---- candidate edges (gen_edges.py:44-90): same lateral offset +- what the longitudinal distance allows ------------------
    rl_coeff = _periodic_spline_coeffs(raceline)
    src_g, dst_g, src_l = [], [], []
    for i in range(L):
        j = i + 1
        if j >= L:
            if not closed:
                break
            j -= L
        Ki, Kj = int(nodes_in_layer[i]), int(nodes_in_layer[j])
        a = np.arange(Ki)
        ref = rl_idx[j] + a - rl_idx[i]
        d_end = node_pos[layer_off[j] + np.clip(ref, 0, Kj - 1)]
        d_start = node_pos[layer_off[i] + a]
        dist = np.sqrt(np.power(d_end[:, 0] - d_start[:, 0], 2) + np.power(d_end[:, 1] - d_start[:, 1], 2))
        steps = np.array([int(round(v)) for v in dist * cfg["lat_offset"] / res], dtype=np.int64)     # Python's round (half to even)
        for n in range(Ki):
            ends = np.arange(max(0, ref[n] - steps[n]), min(Kj, ref[n] + steps[n] + 1))
            src_g.append(np.full(ends.size, layer_off[i] + n))
            dst_g.append(layer_off[j] + ends)
            src_l.append(np.full(ends.size, i))
    src_g, dst_g, src_l = np.concatenate(src_g), np.concatenate(dst_g), np.concatenate(src_l)
    dst_l = np.where(src_l + 1 >= L, src_l + 1 - L, src_l + 1)
    is_rl = ((src_g - layer_off[src_l]) == rl_idx[src_l]) & ((dst_g - layer_off[dst_l]) == rl_idx[dst_l])
    given = np.where(is_rl[:, None], rl_coeff[src_l], 0.0)
    start = np.column_stack((node_pos[src_g], node_psi[src_g]))
    end = np.column_stack((node_pos[dst_g], node_psi[dst_g]))
    min_turn = np.power(vel_l[src_l] * cfg["min_vel_race"], 2) / 10.0
    chord = np.hypot(end[:, 0] - start[:, 0], end[:, 1] - start[:, 1])
    cap = int(math.ceil(1.7 * float(chord.max()) / cfg["stepsize_approx"])) + 3
    ev = evaluate_edges(start, end, 1.0 / min_turn, is_rl.astype(np.int32), given, cfg["stepsize_approx"], 1.0 / cfg["veh_turn"], cap)
    if np.any(ev["n_samples"] > cap):
        raise RuntimeError("edge with more samples than the staging capacity")
    alive = ev["valid"].astype(bool)

    # ---- dead ends (prune_graph.py:8-72): drop edges into nodes without children / out of nodes without parents, to a fixed
    #      point; on open tracks the first and the last layer are exempt -------------------------------------------------------
    node_layer = np.repeat(np.arange(L), nodes_in_layer)
    exempt = np.zeros(V, bool) if closed else ((node_layer == 0) | (node_layer == L - 1))
    while True:
        outdeg = np.bincount(src_g[alive], minlength=V)
        indeg = np.bincount(dst_g[alive], minlength=V)
        dead = alive & (((outdeg[dst_g] == 0) & ~exempt[dst_g] & (indeg[dst_g] > 0)) | ((indeg[src_g] == 0) & ~exempt[src_g] & (outdeg[src_g] > 0)))
        if not dead.any():
            break
        alive &= ~dead

    # ---- offline cost (gen_offline_cost.py:12-82) ------------------------------------------------------------------------------
    ln = ev["length"]
    rl_dist = np.abs(rl_idx[dst_l] - (dst_g - layer_off[dst_l])) * res
    cost = (cfg["w_curv_avg"] * np.power(ev["kappa_avg"], 2) * ln + cfg["w_curv_peak"] * np.power(ev["kappa_range"], 2) * ln
            + cfg["w_length"] * ln + np.minimum(cfg["w_raceline"] * ln * rl_dist, cfg["w_raceline_sat"] * ln))

    # ---- assembly: CSC by destination node, in-edges sorted by source node ------------------------------------------------------
    keep = np.flatnonzero(alive)
    order = keep[np.lexsort((src_g[keep], dst_g[keep]))]
    in_ptr = np.concatenate(([0], np.cumsum(np.bincount(dst_g[order], minlength=V))))
    ns = ev["n_samples"][order].astype(np.int64)
    samp_ptr = np.concatenate(([0], np.cumsum(ns)))
    samples = np.concatenate([ev["samples"][e, :n, :] for e, n in zip(order, ns)], axis=0) if order.size else np.zeros((0, 5))
    vgoal = np.abs(rl_idx[node_layer] - (np.arange(V) - layer_off[node_layer])) * res * cfg["w_virt_goal"]
    if not cfg.get("virt_goal_n", True):
        # no virtual goal nodes (params/ltpl_config_offline.ini:25): GraphBase.search_graph_layer tries the end layer's nodes in a fixed
        # order instead (GraphBase.py:896-927); goal costs that reproduce that choice: lattice.goal_order_cost
        from .lattice import goal_order_cost
        vgoal = goal_order_cost(rl_idx, nodes_in_layer)
    return Lattice(num_layers=L, lat_resolution=res, lat_offset=cfg["lat_offset"], veh_width=cfg["veh_width"],
                   veh_length=cfg["veh_length"], sampled_resolution=cfg["stepsize_approx"], vel_decrease_lat=cfg["vel_decrease_lat"],
                   min_plan_horizon=cfg["min_plan_horizon"], plan_horizon_mode=cfg["plan_horizon_mode"], closed=closed,
                   virt_goal_node_cost=cfg["w_virt_goal"], nodes_in_layer=nodes_in_layer, raceline_index=rl_idx, s_raceline=s_layers,
                   refline=refline, raceline=raceline, vel_raceline=vel_l, normvec=normvec, track_width_right=w_r,
                   track_width_left=w_l, alpha=alpha, node_pos=node_pos, node_psi=node_psi, vgoal_cost=vgoal, in_ptr=in_ptr,
                   edge_src=(src_g[order] - layer_off[src_l[order]]), edge_cost=cost[order], edge_len=ln[order],
                   edge_coeff=ev["coeff"][order], samp_ptr=samp_ptr, samples=samples, glob_rl=glob_rl)
