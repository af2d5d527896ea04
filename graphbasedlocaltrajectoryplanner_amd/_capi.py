"""
ctypes binding of the C ABI declared in include/ltpl_hip.h (libltpl_hip.so, hand-written HIP for gfx950).

There is deliberately NO CPU fallback: if the shared library is missing or no HIP device is visible, constructing
``HipBackend`` raises. The struct layouts below mirror include/ltpl_hip.h field by field.
"""

import array
import ctypes as C
import os
import numpy as np

from .lattice import Lattice

MAX_ACTIONS = 3
MAX_LAST_NODES = 8
ACT_STRAIGHT, ACT_FOLLOW, ACT_LEFT, ACT_RIGHT, ACT_NONE, ACT_EMERGENCY = 0, 1, 2, 3, -1, 4
ACTION_NAMES = {ACT_STRAIGHT: "straight", ACT_FOLLOW: "follow", ACT_LEFT: "left", ACT_RIGHT: "right"}
PLANNER_MAX_KEYS = 4
ACTION_IDS = {v: k for k, v in ACTION_NAMES.items()}
FLAG_ACTION_SETS, FLAG_OBJ_IN_CONST, FLAG_OBJ_BESIDES, FLAG_HAS_PSI_S = 1, 2, 4, 8
VEL_FB, VEL_BRAKE, VEL_FOLLOW = 0, 1, 2

_STATUS = {0: "OK", 1: "invalid argument", 2: "no HIP device", 3: "HIP runtime error", 4: "capacity exceeded",
           5: "unsupported configuration", 6: "internal error (C++ exception caught at the ABI)"}

_pi32 = C.POINTER(C.c_int32)
_pf64 = C.POINTER(C.c_double)


class LatticeDesc(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("num_nodes", C.c_int32), ("num_edges", C.c_int32),
                ("num_samples", C.c_int32), ("num_glob_rl", C.c_int32), ("closed", C.c_int32),
                ("plan_horizon_mode", C.c_int32), ("reserved0", C.c_int32),
                ("min_plan_horizon", C.c_double), ("lat_resolution", C.c_double), ("lat_offset", C.c_double),
                ("veh_width", C.c_double), ("veh_length", C.c_double), ("sampled_resolution", C.c_double),
                ("vel_decrease_lat", C.c_double),
                ("layer_node_off", _pi32), ("raceline_index", _pi32), ("s_raceline", _pf64), ("refline_x", _pf64),
                ("refline_y", _pf64), ("vel_raceline", _pf64),
                ("node_x", _pf64), ("node_y", _pf64), ("vgoal_cost", _pf64),
                ("in_ptr", _pi32), ("edge_src", _pi32), ("edge_cost", _pf64), ("edge_len", _pf64),
                ("samp_ptr", _pi32),
                ("samp_x", _pf64), ("samp_y", _pf64), ("samp_psi", _pf64), ("samp_len", _pf64),
                ("glob_rl", _pf64),
                ("normvec_x", _pf64), ("normvec_y", _pf64), ("width_right", _pf64), ("width_left", _pf64),
                ("raceline_x", _pf64), ("raceline_y", _pf64), ("node_psi", _pf64)]


class Caps(C.Structure):
    _fields_ = [("max_path_nodes", C.c_int32), ("max_path_pts", C.c_int32), ("max_horizon_edges", C.c_int32),
                ("device", C.c_int32), ("num_cus", C.c_int32), ("lds_bytes_paths", C.c_int32)]


# The per-call INPUT structs declare their pointer members as plain addresses (c_void_p) and are filled with
# ndarray.ctypes.data: a typed ctypes pointer per array (ndarray.ctypes.data_as) costs ~4 us, and a tick packs ~25 of them.
_vp = C.c_void_p


class PathsIn(C.Structure):
    _fields_ = [("n_scen", C.c_int32), ("n_w_last", C.c_int32), ("w_last_edges", _vp),
                ("start_layer", _vp), ("start_node", _vp), ("flags", _vp), ("last_action", _vp),
                ("const_closest", _vp), ("psi_s", _vp),
                ("veh_off", _vp), ("pos_off", _vp), ("veh_radius", _vp), ("pos_x", _vp), ("pos_y", _vp),
                ("zone_off", _vp), ("zone_gid", _vp),
                ("n_last", _vp), ("last_layer", _vp), ("last_node", _vp)]


class PathsOut(C.Structure):
    _fields_ = [("cap_nodes", C.c_int32), ("cap_pts", C.c_int32),
                ("end_layer", _pi32), ("closest_obj_index", _pi32), ("closest_obj_node", _pi32),
                ("n_actions", _pi32),
                ("action_id", _pi32), ("valid", _pi32), ("reduced", _pi32), ("goal_layer", _pi32),
                ("n_nodes", _pi32), ("n_pts", _pi32), ("n_ties", _pi32), ("nodes", _pi32), ("node_idx", _pi32),
                ("coeff", _pf64), ("path_param", _pf64)]


class VelParams(C.Structure):
    _fields_ = [("dyn_model_exp", C.c_double), ("drag_coeff", C.c_double), ("m_veh", C.c_double),
                ("len_veh", C.c_double), ("v_max", C.c_double),
                ("n_ax_max_machines", C.c_int32), ("follow_control_type", C.c_int32),
                ("ax_max_machines", _pf64),
                ("c_p", C.c_double), ("k_p", C.c_double), ("k_d", C.c_double), ("tan_w", C.c_double)]


class VelJob(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n", C.c_int32), ("n_el", C.c_int32), ("has_v_end", C.c_int32),
                ("kappa", _pf64), ("el_lengths", _pf64), ("loc_gg", _pf64),
                ("v_start", C.c_double), ("v_end", C.c_double),
                ("v_ego", C.c_double), ("v_obj", C.c_double), ("safety_d", C.c_double), ("obj_dist", C.c_double),
                ("obj_x", C.c_double), ("obj_y", C.c_double)]


class VelResult(C.Structure):
    _fields_ = [("vx", _pf64), ("too_close", C.c_int32), ("vel_bound", C.c_int32)]


class TickVelIn(C.Structure):
    _fields_ = [("params", C.POINTER(VelParams)), ("gg_ax", C.c_double), ("gg_ay", C.c_double),
                ("gg_brake_scale", C.c_double), ("safety_d", C.c_double), ("v_max_offset", C.c_double),
                ("vel_plan", _vp), ("vel_est", _vp), ("pos_est_x", _vp), ("pos_est_y", _vp),
                ("veh_vel", _vp)]


class ObjectsIn(C.Structure):              # pointer members as plain addresses (see planner.py: a typed pointer costs ~4 us to build)
    _fields_ = [("n_obj", C.c_int32), ("reserved0", C.c_int32), ("dt", C.c_double),
                ("x", C.c_void_p), ("y", C.c_void_p), ("theta", C.c_void_p), ("v", C.c_void_p), ("length", C.c_void_p)]


class ObjectsOut(C.Structure):
    _fields_ = [("on_track", C.c_void_p), ("pred_x", C.c_void_p), ("pred_y", C.c_void_p), ("radius", C.c_void_p)]


def make_objects(x, y, theta, v, length, dt=0.2):
    """(ObjectsIn, ObjectsOut, output arrays dict, keep-alive list) for n flat objects: one float64 array [inputs | outputs], one
    int32 array, views and pointer arithmetic."""
    cols = [np.asarray(a, dtype=np.float64).reshape(-1) for a in (x, y, theta, v, length)]
    n = cols[0].size
    m = max(n, 1)
    buf = np.zeros(8 * m)
    for k, c in enumerate(cols):
        if n:
            buf[k * m:k * m + n] = c
    flags = np.zeros(m, np.int32)
    i, out = ObjectsIn(), ObjectsOut()
    i.n_obj, i.dt = n, float(dt)
    b = buf.ctypes.data
    i.x, i.y, i.theta, i.v, i.length = b, b + 8 * m, b + 16 * m, b + 24 * m, b + 32 * m
    out.pred_x, out.pred_y, out.radius = b + 40 * m, b + 48 * m, b + 56 * m
    out.on_track = flags.ctypes.data
    o = {"on_track": flags[:n], "pred_x": buf[5 * m:5 * m + n], "pred_y": buf[6 * m:6 * m + n], "radius": buf[7 * m:7 * m + n]}
    return i, out, o, [buf, flags]


class TrajOut(C.Structure):
    _fields_ = [("max_rows", C.c_int32), ("reserved0", C.c_int32), ("capacity_rows", C.c_int64),
                ("action_id", _pi32), ("n_rows", _pi32), ("vel_bound", _pi32), ("reduced", _pi32),
                ("row_off", C.POINTER(C.c_int64)), ("rows", _pf64), ("total_rows", C.c_int64)]


class CompactTrajectories(object):
    """Caller buffers of ``ltpl_tick_batch_compact``: packed rows [s, x, y, psi, kappa, vx, ax] in page-locked host memory
    (``ltpl_host_alloc``) + per-slot tables. ``trajectories(s)`` rebuilds the dict Graph_LTPL.calc_vel_profile returns."""

    def __init__(self, lib, n_scen, capacity_rows, max_rows=115):
        self.lib, self.n_scen, self.capacity_rows = lib, int(n_scen), int(capacity_rows)
        n_slots = self.n_scen * MAX_ACTIONS
        self.action_id = np.zeros(n_slots, np.int32)
        self.n_rows = np.zeros(n_slots, np.int32)
        self.vel_bound = np.zeros(n_slots, np.int32)
        self.reduced = np.zeros(n_slots, np.int32)
        self.row_off = np.zeros(n_slots, np.int64)
        lib.ltpl_host_alloc.restype = C.c_void_p
        lib.ltpl_host_alloc.argtypes = [C.c_size_t]
        lib.ltpl_host_free.argtypes = [C.c_void_p]
        self._ptr = lib.ltpl_host_alloc(self.capacity_rows * 7 * 8)
        if not self._ptr:
            raise BackendError("ltpl_host_alloc failed")
        self.rows = np.ctypeslib.as_array((C.c_double * (self.capacity_rows * 7)).from_address(self._ptr)).reshape(-1, 7)
        s = self.struct = TrajOut()
        s.max_rows, s.capacity_rows = int(max_rows), self.capacity_rows
        s.action_id, s.n_rows, s.vel_bound, s.reduced = (_p(a, _pi32) for a in (self.action_id, self.n_rows, self.vel_bound,
                                                                                  self.reduced))
        s.row_off = self.row_off.ctypes.data_as(C.POINTER(C.c_int64))
        s.rows = C.cast(self._ptr, _pf64)

    def trajectories(self, scen, copy=True):
        """The trajectory set of scenario ``scen`` as Graph_LTPL.calc_vel_profile returns it ({action id: [ndarray (rows, 7)]}).
        ``copy=True`` (default): independent arrays -- the reference's callers keep the dict (Graph_LTPL stores it as its action set),
        while the packed buffer is overwritten by the next ltpl_tick_batch_compact and released with this object. ``copy=False``:
        zero-copy VIEWS into the page-locked buffer, only valid until the next call on, or the release of, this object."""
        out = {}
        for a in range(MAX_ACTIONS):
            k = scen * MAX_ACTIONS + a
            if self.n_rows[k] > 0:
                o = int(self.row_off[k])
                rows = self.rows[o:o + int(self.n_rows[k])]
                out[ACTION_NAMES[int(self.action_id[k])]] = [rows.copy() if copy else rows]
        return out

    def __del__(self):
        try:
            if self._ptr:
                self.rows = None
                self.lib.ltpl_host_free(self._ptr)
                self._ptr = None
        except Exception:
            pass


class TickVelOut(C.Structure):
    _fields_ = [("vx", _pf64), ("ax", _pf64), ("vel_bound", _pi32), ("too_close", _pi32)]


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


class LatticeBinding(object):
    """Keeps the column arrays alive that a ``ltpl_lattice_desc`` points to."""

    def __init__(self, lat: Lattice):
        self.lat = lat
        k = self.keep = {}
        k["layer_node_off"] = _i32(lat.layer_off)
        k["raceline_index"] = _i32(lat.raceline_index)
        k["s_raceline"] = _f64(lat.s_raceline)
        k["refline_x"] = _f64(lat.refline[:, 0])
        k["refline_y"] = _f64(lat.refline[:, 1])
        k["vel_raceline"] = _f64(lat.vel_raceline)
        k["node_x"] = _f64(lat.node_pos[:, 0])
        k["node_y"] = _f64(lat.node_pos[:, 1])
        k["vgoal_cost"] = _f64(lat.vgoal_cost)
        k["in_ptr"] = _i32(lat.in_ptr)
        k["edge_src"] = _i32(lat.edge_src)
        k["edge_cost"] = _f64(lat.edge_cost)
        k["edge_len"] = _f64(lat.edge_len)
        k["samp_ptr"] = _i32(lat.samp_ptr)
        k["samp_x"] = _f64(lat.samples[:, 0])
        k["samp_y"] = _f64(lat.samples[:, 1])
        k["samp_psi"] = _f64(lat.samples[:, 2])
        k["samp_len"] = _f64(lat.samples[:, 4])
        k["glob_rl"] = _f64(lat.glob_rl)
        k["normvec_x"] = _f64(lat.normvec[:, 0])
        k["normvec_y"] = _f64(lat.normvec[:, 1])
        k["width_right"] = _f64(lat.track_width_right)
        k["width_left"] = _f64(lat.track_width_left)
        k["raceline_x"] = _f64(lat.raceline[:, 0])
        k["raceline_y"] = _f64(lat.raceline[:, 1])
        k["node_psi"] = _f64(lat.node_psi)
        d = self.desc = LatticeDesc()
        d.num_layers, d.num_nodes, d.num_edges = lat.num_layers, lat.num_nodes, lat.num_edges
        d.num_samples, d.num_glob_rl = lat.num_samples, lat.glob_rl.shape[0]
        d.closed = int(lat.closed)
        d.plan_horizon_mode = {"distance": 0, "layers": 1}[lat.plan_horizon_mode]
        d.min_plan_horizon = lat.min_plan_horizon
        d.lat_resolution, d.lat_offset = lat.lat_resolution, lat.lat_offset
        d.veh_width, d.veh_length = lat.veh_width, lat.veh_length
        d.sampled_resolution, d.vel_decrease_lat = lat.sampled_resolution, lat.vel_decrease_lat
        for name, arr in k.items():
            setattr(d, name, _p(arr, _pi32 if arr.dtype == np.int32 else _pf64))


class VelParamSet(object):
    """VpForwardBackward constructor + update_dyn_parameters state (VpForwardBackward.py:22-84)."""

    def __init__(self, dyn_model_exp=1.0, drag_coeff=0.85, m_veh=1000.0, len_veh=4.7, v_max=100.0,
                 ax_max_machines=((100.0, 5.0),), follow_control_type="PD", follow_control_params=None):
        cp = follow_control_params or {"c_p": 1.25, "k_d": 0.025, "k_p": 0.2}
        self.axm = _f64(np.atleast_2d(ax_max_machines))
        if self.axm.ndim != 2 or self.axm.shape[1] != 2:
            raise RuntimeError("ax_max_machines must consist of the two columns [vx, ax_max_machines]!")
        s = self.struct = VelParams()
        s.dyn_model_exp, s.drag_coeff, s.m_veh, s.len_veh, s.v_max = dyn_model_exp, drag_coeff, m_veh, len_veh, v_max
        s.n_ax_max_machines = self.axm.shape[0]
        if follow_control_type not in ("PD", "PDtan"):
            raise ValueError('Unsupported control type "' + str(follow_control_type) + '"!')
        s.follow_control_type = 0 if follow_control_type == "PD" else 1
        s.ax_max_machines = _p(self.axm, _pf64)
        s.c_p, s.k_p, s.k_d = cp["c_p"], cp["k_p"], cp["k_d"]
        s.tan_w = cp.get("tan_w", 1.0)


class PathsBatch(object):
    """
    Host-side packing of n scenarios into ``ltpl_paths_in``. A scenario is a dict with keys
      start_node (layer, node); action_sets (bool); obj_in_const, obj_besides (bool); last_action (name | None);
      const_closest (int | None); psi_s (float | None); vehicles [(radius, positions (k, 2) own pos first)];
      zone_gids (iterable of global node ids); last_nodes [[layer, node], ...] | None
    ``w_last_edges`` is shared by the batch (it is a config value: OnlineTrajectoryHandler.py:109).
    """

    def __init__(self, scenarios, w_last_edges=()):
        # Everything is collected in plain Python lists and converted ONCE per element type; the public attributes are views
        # into those two arrays and the struct's pointer members are base address + offset. (A NumPy operation per field and
        # vehicle plus an address lookup per column made the packing of ONE scenario cost as much as its planning kernel.)
        n = len(scenarios)
        self.n_scen = n
        self.n_w_last = len(w_last_edges)
        if self.n_w_last > MAX_LAST_NODES - 1:
            raise ValueError("w_last_edges longer than %d entries is not supported" % (MAX_LAST_NODES - 1))
        M = MAX_LAST_NODES
        start_layer, start_node, flags, last_action, const_closest, psi_s, n_last = [], [], [], [], [], [], []
        last_layer, last_node = [], []
        veh_off, pos_off, radius, px, py, zone_off, zone = [0], [0], [], [], [], [0], []
        for sc in scenarios:
            sn = sc["start_node"]
            start_layer.append(int(sn[0])); start_node.append(int(sn[1]))
            f = 0
            if sc.get("action_sets", True):
                f |= FLAG_ACTION_SETS
            if sc.get("obj_in_const", False):
                f |= FLAG_OBJ_IN_CONST
            if sc.get("obj_besides", False):
                f |= FLAG_OBJ_BESIDES
            ps = sc.get("psi_s")
            if ps is not None:
                f |= FLAG_HAS_PSI_S
            psi_s.append(0.0 if ps is None else float(ps))
            flags.append(f)
            la = sc.get("last_action")
            last_action.append(ACTION_IDS.get(la, ACT_NONE) if isinstance(la, str) else ACT_NONE)
            cc = sc.get("const_closest")
            const_closest.append(-1 if cc is None else int(cc))
            for radius_k, positions in sc.get("vehicles", ()):
                radius.append(float(radius_k))
                if isinstance(positions, np.ndarray):
                    positions = positions.reshape(-1, 2).tolist()
                elif len(positions) and not isinstance(positions[0], (list, tuple, np.ndarray)):
                    positions = [positions]                     # a single (x, y)
                for xy in positions:
                    px.append(float(xy[0])); py.append(float(xy[1]))
                pos_off.append(len(px))
            veh_off.append(len(radius))
            zg = sc.get("zone_gids", ())
            zone.extend(zg.tolist() if isinstance(zg, np.ndarray) else [int(g) for g in zg])
            zone_off.append(len(zone))
            ll, lnn = [-1] * M, [-1] * M
            k = 0
            ln = sc.get("last_nodes")
            if ln:
                # only nodes that can take part in a cost discount are shipped (w_last_edges has <= MAX-1 entries)
                for node in ln[:M]:
                    if node is None or node[0] is None or node[1] is None:
                        break
                    ll[k], lnn[k] = int(node[0]), int(node[1])
                    k += 1
            n_last.append(k)
            last_layer += ll; last_node += lnn
        self.n_veh_total = len(radius)
        if not radius:
            radius = [0.0]
        if not px:
            px, py = [0.0], [0.0]
        if not zone:
            zone = [0]
        wl = [float(w) for w in w_last_edges] if self.n_w_last else [0.0]
        # (array.array: list -> C buffer and its address in 1.5 us; np.array + ndarray.ctypes.data take 3 us per array)
        ints = array.array("i", start_layer + start_node + flags + last_action + const_closest + n_last + last_layer + last_node +
                           veh_off + pos_off + zone_off + zone)
        dbls = array.array("d", wl + psi_s + radius + px + py)
        self._ints, self._dbls = ints, dbls
        bi, bd = ints.buffer_info()[0], dbls.buffer_info()[0]
        st = self.struct = PathsIn()
        st.n_scen, st.n_w_last = n, self.n_w_last
        # pointer members = base address + offset, in the order the lists were concatenated; the array attributes of the same names
        # (start_layer, veh_off, pos_x, ...) are views created on first access (__getattr__) from the element counts kept here
        nvo, npo, nzo, nz, nw, nr, npx = len(veh_off), len(pos_off), len(zone_off), len(zone), len(wl), len(radius), len(px)
        self._counts = (n, n, n, n, n, n, n * M, n * M, nvo, npo, nzo, nz), (nw, n, nr, npx, npx)
        st.start_layer = bi; bi += 4 * n
        st.start_node = bi; bi += 4 * n
        st.flags = bi; bi += 4 * n
        st.last_action = bi; bi += 4 * n
        st.const_closest = bi; bi += 4 * n
        st.n_last = bi; bi += 4 * n
        st.last_layer = bi; bi += 4 * n * M
        st.last_node = bi; bi += 4 * n * M
        st.veh_off = bi; bi += 4 * nvo
        st.pos_off = bi; bi += 4 * npo
        st.zone_off = bi; bi += 4 * nzo
        st.zone_gid = bi
        st.w_last_edges = bd; bd += 8 * nw
        st.psi_s = bd; bd += 8 * n
        st.veh_radius = bd; bd += 8 * nr
        st.pos_x = bd; bd += 8 * npx
        st.pos_y = bd

    _INT_VIEWS = ("start_layer", "start_node", "flags", "last_action", "const_closest", "n_last", "last_layer", "last_node",
                  "veh_off", "pos_off", "zone_off", "zone_gid")
    _DBL_VIEWS = ("w_last", "psi_s", "veh_radius", "pos_x", "pos_y")

    def __getattr__(self, name):
        # only reached for names that are not instance attributes yet: materialise the view once
        counts = self.__dict__.get("_counts")
        if counts is None:
            raise AttributeError(name)
        if name in self._INT_VIEWS:
            k, cnt, arr = self._INT_VIEWS.index(name), counts[0], np.frombuffer(self._ints, np.int32)
        elif name in self._DBL_VIEWS:
            k, cnt, arr = self._DBL_VIEWS.index(name), counts[1], np.frombuffer(self._dbls, np.float64)
        else:
            raise AttributeError(name)
        o = sum(cnt[:k])
        v = arr[o:o + cnt[k]]
        if name in ("last_layer", "last_node"):
            v = v.reshape(self.n_scen, MAX_LAST_NODES)
        self.__dict__[name] = v
        return v


class PathsResult(object):
    """Caller-allocated output buffers of seam (1) plus accessors that rebuild the reference's Python structures."""

    def __init__(self, n_scen, cap_nodes, cap_pts):
        self.n_scen, self.cap_nodes, self.cap_pts = n_scen, int(cap_nodes), int(cap_pts)
        A = MAX_ACTIONS
        self.end_layer = np.zeros(n_scen, np.int32)
        self.closest_obj_index = np.zeros(n_scen, np.int32)
        self.closest_obj_node = np.zeros((n_scen, 2), np.int32)
        self.n_actions = np.zeros(n_scen, np.int32)
        for name in ("action_id", "valid", "reduced", "goal_layer", "n_nodes", "n_pts", "n_ties"):
            setattr(self, name, np.zeros((n_scen, A), np.int32))
        self.nodes = np.zeros((n_scen, A, self.cap_nodes), np.int32)
        self.node_idx = np.zeros((n_scen, A, self.cap_nodes), np.int32)
        self.coeff = np.zeros((n_scen, A, self.cap_nodes, 8))
        self.path_param = np.zeros((n_scen, A, self.cap_pts, 5))
        s = self.struct = PathsOut()
        s.cap_nodes, s.cap_pts = self.cap_nodes, self.cap_pts
        for name in ("end_layer", "closest_obj_index", "closest_obj_node", "n_actions", "action_id", "valid",
                     "reduced", "goal_layer", "n_nodes", "n_pts", "n_ties", "nodes", "node_idx"):
            setattr(s, name, _p(getattr(self, name), _pi32))
        s.coeff, s.path_param = _p(self.coeff, _pf64), _p(self.path_param, _pf64)

    def action_sets(self, s, start_layer, num_layers):
        """
        The 6-tuple of main_online_path_gen (main_online_path_gen.py:333-334) for scenario ``s``: dicts keyed by action
        name in insertion order, each value a one-element list; absent keys = no solution.
        """
        nodes, node_idx, coeff, path_param, red_len = {}, {}, {}, {}, {}
        start_layer = int(start_layer)
        for a in range(int(self.n_actions[s])):
            if not self.valid[s, a]:
                continue
            name = ACTION_NAMES[int(self.action_id[s, a])]
            nn, npts = int(self.n_nodes[s, a]), int(self.n_pts[s, a])
            nl = self.nodes[s, a, :nn].tolist()                 # (one conversion instead of a NumPy scalar access per node)
            nodes[name] = [[[(start_layer + i) % num_layers, nl[i]] for i in range(nn)]]
            node_idx[name] = [self.node_idx[s, a, :nn].tolist()]
            coeff[name] = [self.coeff[s, a, :nn - 1, :].copy()]
            path_param[name] = [self.path_param[s, a, :npts, :].copy()]
            red_len[name] = [bool(self.reduced[s, a])]
        coi = int(self.closest_obj_index[s])
        return nodes, node_idx, coeff, path_param, red_len, (None if coi < 0 else coi)


class TickVelBatch(object):
    def __init__(self, params: VelParamSet, n_scen, vel_plan, vel_est, pos_est, veh_vel, gg=(5.0, 5.0),
                 gg_brake_scale=1.0, safety_d=30.0, v_max_offset=0.1):
        # one array, views and pointer arithmetic (see PathsBatch)
        self.params = params
        n = int(n_scen)
        pos = np.asarray(pos_est, dtype=np.float64).reshape(n, 2)
        vv = np.asarray(veh_vel, dtype=np.float64).reshape(-1)
        nv = max(len(vv), 1)
        buf = np.zeros(4 * n + nv)
        buf[0:n] = vel_plan
        buf[n:2 * n] = vel_est
        buf[2 * n:3 * n] = pos[:, 0]
        buf[3 * n:4 * n] = pos[:, 1]
        if len(vv):
            buf[4 * n:] = vv
        self._buf = buf
        self.vel_plan, self.vel_est = buf[0:n], buf[n:2 * n]
        self.pos_x, self.pos_y, self.veh_vel = buf[2 * n:3 * n], buf[3 * n:4 * n], buf[4 * n:]
        s = self.struct = TickVelIn()
        s.params = C.pointer(params.struct)
        s.gg_ax, s.gg_ay, s.gg_brake_scale = float(gg[0]), float(gg[1]), float(gg_brake_scale)
        s.safety_d, s.v_max_offset = float(safety_d), float(v_max_offset)
        b = buf.ctypes.data
        s.vel_plan, s.vel_est, s.pos_est_x, s.pos_est_y, s.veh_vel = b, b + 8 * n, b + 16 * n, b + 24 * n, b + 32 * n


class TickVelResult(object):
    def __init__(self, n_scen, cap_pts):
        A = MAX_ACTIONS
        self.vx = np.zeros((n_scen, A, cap_pts))
        self.ax = np.zeros((n_scen, A, cap_pts))
        self.vel_bound = np.zeros((n_scen, A), np.int32)
        self.too_close = np.zeros((n_scen, A), np.int32)
        s = self.struct = TickVelOut()
        s.vx, s.ax = _p(self.vx, _pf64), _p(self.ax, _pf64)
        s.vel_bound, s.too_close = _p(self.vel_bound, _pi32), _p(self.too_close, _pi32)


def make_vel_jobs(jobs):
    """
    jobs: list of dicts {mode, kappa, el_lengths, loc_gg, v_start, v_end(optional), + follow fields}. Returns
    (ctypes array of VelJob, ctypes array of VelResult, list of output arrays, keep-alive list).
    """
    n = len(jobs)
    jarr = (VelJob * n)()
    rarr = (VelResult * n)()
    outs, keep = [], []
    for i, jb in enumerate(jobs):
        kappa, el, gg = _f64(jb["kappa"]), _f64(jb["el_lengths"]), _f64(jb["loc_gg"])
        if gg.ndim != 2 or gg.shape[1] != 2:
            raise RuntimeError("loc_gg must consist of two columns: [ax_max, ay_max]!")
        if gg.shape[0] != kappa.size:
            raise RuntimeError("Length of loc_gg and kappa must be equal!")
        if el.size == 0:
            el = _f64([0.0])[:0]
        out = np.zeros(kappa.size)
        keep.extend((kappa, el, gg))
        outs.append(out)
        j = jarr[i]
        j.mode, j.n, j.n_el = int(jb["mode"]), kappa.size, int(jb["el_lengths"].size if hasattr(jb["el_lengths"], "size")
                                                                else len(jb["el_lengths"]))
        j.has_v_end = int(jb.get("v_end") is not None)
        j.kappa, j.el_lengths, j.loc_gg = _p(kappa, _pf64), _p(el, _pf64), _p(gg, _pf64)
        j.v_start = float(jb["v_start"])
        j.v_end = float(jb["v_end"]) if jb.get("v_end") is not None else 0.0
        for key in ("v_ego", "v_obj", "safety_d", "obj_dist"):
            setattr(j, key, float(jb.get(key, 0.0)))
        op = jb.get("obj_pos", (0.0, 0.0))
        j.obj_x, j.obj_y = float(op[0]), float(op[1])
        rarr[i].vx = _p(out, _pf64)
    return jarr, rarr, outs, keep


def pack_const_segment_args(const_path_seg, pos_est, vehicles):
    """ctypes arguments (seg, n_rows, pos_est, n_veh, x, y, radius) of ltpl_const_segment_test + keep-alive list."""
    seg = None if const_path_seg is None else _f64(np.asarray(const_path_seg, dtype=np.float64)[:, :5])
    n_rows = 0 if seg is None else seg.shape[0]
    pos = None if pos_est is None else _f64(np.asarray(pos_est, dtype=np.float64).reshape(-1)[:2])
    vx = _f64([float(p[0][0]) for _, p in vehicles] or [0.0])
    vy = _f64([float(p[0][1]) for _, p in vehicles] or [0.0])
    vr = _f64([float(r) for r, _ in vehicles] or [0.0])
    return (None if seg is None else _p(seg, _pf64), n_rows, None if pos is None else _p(pos, _pf64), len(vehicles),
            _p(vx, _pf64), _p(vy, _pf64), _p(vr, _pf64), None, (seg, pos, vx, vy, vr))


def default_library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libltpl_hip.so")


def experiment_library_path():
    """The -DLTPL_EXPERIMENT build (timing / fault-injection switches, include/ltpl_hip.h): tools/ and one fault-injection test."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libltpl_hip_exp.so")


class BackendError(RuntimeError):
    pass


CREATE_PERSISTENT_TICK = 1      # LTPL_CREATE_PERSISTENT_TICK


class PersistentStats(C.Structure):        # ltpl_persistent_stats
    _fields_ = [("enabled", C.c_int32), ("resident", C.c_int32), ("ticks", C.c_int64), ("launches", C.c_int64),
                ("device_us_mean", C.c_double), ("device_us_last", C.c_double), ("idle_ms", C.c_double)]


class HipBackend(object):
    """One handle = one lattice resident in the HBM of one MI355X. Raises if the HIP library / device is missing."""

    def __init__(self, lattice: Lattice, device: int = -1, lib_path: str = None, persistent_tick: bool = False):
        """``persistent_tick``: ltpl_create_ex(LTPL_CREATE_PERSISTENT_TICK) -- single-scenario ``tick_batch`` calls are served by a resident
        kernel behind a mailbox in page-locked memory (include/ltpl_hip.h; ``persistent_stats()`` tells whether the lattice engages it)."""
        path = lib_path or os.environ.get("LTPL_HIP_LIB") or default_library_path()
        if not os.path.isfile(path):
            raise BackendError("libltpl_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'`; there is no CPU fallback" % path)
        self.lib = C.CDLL(path)
        self.lib_path = os.path.abspath(path)          # (LTPL_HIP_LIB redirects the product library: bench.py prints the resolved path + hashes)
        self._declare()
        self.binding = LatticeBinding(lattice)
        self.lattice = lattice
        self.handle = C.c_void_p()
        if persistent_tick:
            if not hasattr(self.lib, "ltpl_create_ex"):
                raise BackendError("this libltpl_hip.so predates ltpl_create_ex (ABI v9): no persistent tick")
            self.lib.ltpl_create_ex.argtypes = [C.POINTER(LatticeDesc), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
            rc = self.lib.ltpl_create_ex(C.byref(self.binding.desc), int(device), CREATE_PERSISTENT_TICK, C.byref(self.handle))
        else:
            rc = self.lib.ltpl_create(C.byref(self.binding.desc), int(device), C.byref(self.handle))
        if rc != 0:
            msg = self.lib.ltpl_last_error(None)
            raise BackendError("ltpl_create failed (%s): %s" % (_STATUS.get(rc, rc), (msg or b"").decode()))
        caps = Caps()
        self._check(self.lib.ltpl_get_caps(self.handle, C.byref(caps)))
        self.caps = caps

    def _declare(self):
        L = self.lib
        L.ltpl_create.argtypes = [C.POINTER(LatticeDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.ltpl_destroy.argtypes = [C.c_void_p]
        L.ltpl_get_caps.argtypes = [C.c_void_p, C.POINTER(Caps)]
        L.ltpl_last_error.argtypes = [C.c_void_p]
        L.ltpl_last_error.restype = C.c_char_p
        L.ltpl_plan_paths.argtypes = [C.c_void_p, C.POINTER(PathsIn), C.POINTER(PathsOut)]
        L.ltpl_plan_paths_mask.argtypes = [C.c_void_p, C.POINTER(PathsIn), C.POINTER(PathsOut), C.c_int32, C.c_void_p]
        L.ltpl_vel_profile.argtypes = [C.c_void_p, C.POINTER(VelParams), C.c_int, C.POINTER(VelJob),
                                       C.POINTER(VelResult)]
        L.ltpl_tick_batch.argtypes = [C.c_void_p, C.POINTER(PathsIn), C.POINTER(TickVelIn), C.POINTER(PathsOut),
                                      C.POINTER(TickVelOut)]
        L.ltpl_tick_batch_compact.argtypes = [C.c_void_p, C.POINTER(PathsIn), C.POINTER(TickVelIn), C.POINTER(TrajOut)]
        L.ltpl_batch_upload.argtypes = [C.c_void_p, C.POINTER(PathsIn), C.POINTER(TickVelIn), C.c_int32, C.c_int32]
        L.ltpl_batch_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.ltpl_batch_download.argtypes = [C.c_void_p, C.POINTER(PathsOut), C.POINTER(TickVelOut)]
        L.ltpl_batch_run_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.ltpl_batch_last_paths_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.ltpl_process_objects.argtypes = [C.c_void_p, C.POINTER(ObjectsIn), C.POINTER(ObjectsOut)]
        L.ltpl_raceline_s.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.ltpl_const_segment_test.argtypes = [C.c_void_p, _pf64, C.c_int32, _pf64, C.c_int32, _pf64, _pf64, _pf64,
                                              _pi32, _pi32]

    def paths_kernel_symbol(self, team_waves=1):
        """Mangled-name prefix of the path kernel this handle launches (batch form: team_waves = 1); '' for a library that predates it."""
        if not hasattr(self.lib, "ltpl_paths_kernel_symbol"):
            return ""
        f = self.lib.ltpl_paths_kernel_symbol
        f.argtypes, f.restype = [C.c_void_p, C.c_int32], C.c_char_p
        return (f(self.handle, int(team_waves)) or b"").decode()

    def persistent_stats(self):
        """ltpl_tick_persistent_stats as a dict (enabled, resident, ticks, launches, device_us_mean, device_us_last, idle_ms)."""
        st = PersistentStats()
        f = self.lib.ltpl_tick_persistent_stats
        f.argtypes = [C.c_void_p, C.POINTER(PersistentStats)]
        self._check(f(self.handle, C.byref(st)))
        return {k: getattr(st, k) for k, _ in PersistentStats._fields_}

    def persistent_stop(self):
        """The resident tick kernel leaves now (it is started again by the next single tick); no-op when none is resident."""
        f = self.lib.ltpl_tick_persistent_stop
        f.argtypes = [C.c_void_p]
        self._check(f(self.handle))

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.ltpl_last_error(self.handle)
            raise BackendError("libltpl_hip: %s: %s" % (_STATUS.get(rc, rc), (msg or b"").decode()))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.ltpl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- seam (1) ----
    def new_paths_result(self, n_scen):
        return PathsResult(n_scen, self.caps.max_path_nodes, self.caps.max_path_pts)

    def plan_paths(self, batch: PathsBatch, result: PathsResult = None) -> PathsResult:
        if result is None:
            result = self.new_paths_result(batch.n_scen)
        self._check(self.lib.ltpl_plan_paths(self.handle, C.byref(batch.struct), C.byref(result.struct)))
        return result

    def plan_paths_mask(self, batch: PathsBatch, team_waves: int = 0):
        """Diagnostics (ltpl_plan_paths_mask): seam (1) plus the obstacle x edge mask as the path kernel computed it,
        uint8 [n_scen, num_edges] by global edge id. ``team_waves``: 0 = automatic, 1 = one-wave batch kernel, 4 = latency kernel."""
        result = self.new_paths_result(batch.n_scen)
        blocked = np.zeros((batch.n_scen, self.lattice.num_edges), np.uint8)
        self._check(self.lib.ltpl_plan_paths_mask(self.handle, C.byref(batch.struct), C.byref(result.struct), int(team_waves),
                                                  blocked.ctypes.data))
        return result, blocked

    def const_segment_test(self, const_path_seg, pos_est, vehicles):
        """(obj_in_const_path, object_besides_const_path, closest object index | None) of main_online_path_gen.py:76-122;
        ``vehicles`` = [(radius, positions (k, 2) own position first)]."""
        args = pack_const_segment_args(const_path_seg, pos_est, vehicles)
        flags, closest = C.c_int32(0), C.c_int32(-1)
        self._check(self.lib.ltpl_const_segment_test(self.handle, *args[:7], C.byref(flags), C.byref(closest)))
        return (bool(flags.value & FLAG_OBJ_IN_CONST), bool(flags.value & FLAG_OBJ_BESIDES),
                None if closest.value < 0 else int(closest.value))

    def raceline_s(self, pos):
        """Global s coordinate of ``pos`` on the race line (get_s_coord with closed=True, Graph_LTPL.py:436-440)."""
        s = C.c_double(0.0)
        self._check(self.lib.ltpl_raceline_s(self.handle, float(pos[0]), float(pos[1]), C.byref(s)))
        return float(s.value)

    # ---- object ingestion ----
    def process_objects(self, x, y, theta, v, length, dt=0.2):
        """on_track flags, constant-velocity prediction points and radii for a flat list of objects."""
        i, o, arrays, keep = make_objects(x, y, theta, v, length, dt)
        if i.n_obj > 0:
            self._check(self.lib.ltpl_process_objects(self.handle, C.byref(i), C.byref(o)))
        return arrays

    # ---- seam (2) ----
    def vel_profile(self, params: VelParamSet, jobs):
        jarr, rarr, outs, keep = make_vel_jobs(jobs)
        self._check(self.lib.ltpl_vel_profile(self.handle, C.byref(params.struct), len(jobs), jarr, rarr))
        return [(outs[i], bool(rarr[i].too_close), bool(rarr[i].vel_bound)) for i in range(len(jobs))]

    # ---- fused tick ----
    def tick_batch(self, batch: PathsBatch, vel: TickVelBatch, result: PathsResult = None,
                   vresult: TickVelResult = None):
        if result is None:
            result = self.new_paths_result(batch.n_scen)
        if vresult is None:
            vresult = TickVelResult(batch.n_scen, result.cap_pts)
        self._check(self.lib.ltpl_tick_batch(self.handle, C.byref(batch.struct), C.byref(vel.struct),
                                             C.byref(result.struct), C.byref(vresult.struct)))
        return result, vresult

    def new_compact_trajectories(self, n_scen, max_rows=115, capacity_rows=None):
        cap = capacity_rows if capacity_rows is not None else n_scen * MAX_ACTIONS * (max_rows if max_rows > 0 else self.caps.max_path_pts)
        return CompactTrajectories(self.lib, n_scen, cap, max_rows)

    def tick_batch_compact(self, batch: PathsBatch, vel: TickVelBatch, out: "CompactTrajectories"):
        """Fused tick with the compact result form: only the trajectory rows in use cross PCIe (ltpl_tick_batch_compact)."""
        self._check(self.lib.ltpl_tick_batch_compact(self.handle, C.byref(batch.struct), C.byref(vel.struct), C.byref(out.struct)))
        return out

    def batch_upload(self, batch: PathsBatch, vel: TickVelBatch):
        self._check(self.lib.ltpl_batch_upload(self.handle, C.byref(batch.struct), C.byref(vel.struct),
                                               self.caps.max_path_nodes, self.caps.max_path_pts))
        self._resident_n = batch.n_scen

    def batch_run(self, reps=1, timed=True):
        ms = C.c_float(0.0)
        self._check(self.lib.ltpl_batch_run(self.handle, int(reps), C.byref(ms) if timed else None))
        return float(ms.value)

    def batch_last_paths_ms(self):
        """Average path-kernel duration inside the last timed batch_run (HIP events around each launch, overlapped run)."""
        ms = C.c_float(0.0)
        self._check(self.lib.ltpl_batch_last_paths_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def batch_run_profile(self, reps=10):
        """ms per launch of (path kernel, follow preparation, velocity lane kernel), HIP events on the library's stream."""
        ms = (C.c_float * 3)()
        self._check(self.lib.ltpl_batch_run_profile(self.handle, int(reps), ms))
        return [float(v) / reps for v in ms]

    def batch_download(self):
        result = self.new_paths_result(self._resident_n)
        vresult = TickVelResult(self._resident_n, result.cap_pts)
        self._check(self.lib.ltpl_batch_download(self.handle, C.byref(result.struct), C.byref(vresult.struct)))
        return result, vresult
