"""
ctypes binding of the planner entry points (``ltpl_planner_*``, include/ltpl_hip.h ABI v3+): the iterative memory of the
reference's ``OnlineTrajectoryHandler`` (graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:24-1040) lives in C++ behind
the ABI, batched over ``n_scen`` independent planners on one lattice handle. One tick is two C calls:

    calc_paths        = Graph_LTPL.calc_paths        (Graph_LTPL.py:300-340)
    calc_vel_profile  = Graph_LTPL.calc_vel_profile  (Graph_LTPL.py:344-408)

The accessors rebuild the reference's Python structures (dicts keyed by action name, one-element lists) from the flat views.
"""
import ctypes as C
import math
import struct
import numpy as np

from . import _capi
from ._capi import _pi32, _pf64, _p, _f64, _i32

K = _capi.PLANNER_MAX_KEYS
KEY_NAMES = dict(_capi.ACTION_NAMES)
KEY_NAMES[_capi.ACT_EMERGENCY] = "emergency"
KEY_IDS = {v: k for k, v in KEY_NAMES.items()}


class PlannerConfig(C.Structure):
    _fields_ = [("n_scen", C.c_int32), ("n_w_last", C.c_int32), ("w_last_edges", _pf64),
                ("v_max_offset", C.c_double), ("delaycomp", C.c_double), ("calc_time_safety", C.c_double),
                ("calc_time_buffer_len", C.c_int32), ("filt_window_width", C.c_int32),
                ("dyn_model_exp", C.c_double), ("drag_coeff", C.c_double), ("m_veh", C.c_double),
                ("follow_control_type", C.c_int32), ("reserved0", C.c_int32),
                ("c_p", C.c_double), ("k_p", C.c_double), ("k_d", C.c_double), ("tan_w", C.c_double)]


# Per-tick input structs: the pointer members are declared as plain addresses (c_void_p) and filled with ndarray.ctypes.data --
# building a typed ctypes pointer per array (ndarray.ctypes.data_as) costs ~4 us each, i.e. more than the C call's own host work.
_vp = C.c_void_p


class PlannerPathsIn(C.Structure):
    _fields_ = [("prev_action", _vp), ("t_now", _vp), ("veh_off", _vp), ("pos_off", _vp),
                ("veh_radius", _vp), ("veh_vel", _vp), ("pos_x", _vp), ("pos_y", _vp),
                ("zone_off", _vp), ("zone_gid", _vp)]


class PlannerVelIn(C.Structure):
    _fields_ = [("pos_est_x", _vp), ("pos_est_y", _vp), ("vel_est", _vp), ("vel_max", _vp),
                ("gg_scale", _vp), ("gg_ax", _vp), ("gg_ay", _vp), ("safety_d", _vp),
                ("incl_emerg_traj", _vp), ("n_ax_max_machines", C.c_int32), ("n_ax_tables", C.c_int32),
                ("ax_max_machines", _vp), ("gg_row_off", _vp), ("gg_rows", _vp),
                ("ax_table_off", _vp), ("ax_table_idx", _vp)]         # ABI v6: machine tables per planner (fleet)


class _Staging(object):
    """Persistent staging memory of the per-tick inputs: one double and one int32 array that grow on demand. The values of a tick are
    written with ONE ``struct.pack_into`` per array and the input struct's pointer members are set to addresses inside -- measured
    against a fresh ndarray + ``ndarray.ctypes.data`` per column (2.3 us a piece) this takes the packing of a tick from 27 to ~8 us."""

    def __init__(self):
        self.cd = self.ci = 0
        self.d = self.i = None
        self.ad = self.ai = 0
        self._grow(256, 256)

    def _grow(self, nd, ni):
        if nd > self.cd:
            self.cd = max(nd, 2 * self.cd)
            self.d = (C.c_double * self.cd)()
            self.ad = C.addressof(self.d)
        if ni > self.ci:
            self.ci = max(ni, 2 * self.ci)
            self.i = (C.c_int32 * self.ci)()
            self.ai = C.addressof(self.i)

    def put(self, doubles, ints):
        self._grow(len(doubles), len(ints))
        struct.pack_into("%dd" % len(doubles), self.d, 0, *doubles)
        struct.pack_into("%di" % len(ints), self.i, 0, *ints)


class PlannerCaps(C.Structure):
    _fields_ = [("cap_rows", C.c_int32), ("cap_nodes", C.c_int32)]


class PathsView(C.Structure):
    _fields_ = [("n_keys", C.c_int32), ("key_id", C.c_int32 * K), ("n_rows", C.c_int32 * K), ("n_nodes", C.c_int32 * K),
                ("red_len", C.c_int32 * K), ("start_node", C.c_int32 * 2), ("const_rows", C.c_int32),
                ("closest_obj_index", C.c_int32),
                ("path_param", _pf64 * K), ("coeff", _pf64 * K), ("nodes", _pi32 * K), ("node_idx", _pi32 * K)]


class TrajView(C.Structure):
    _fields_ = [("n_keys", C.c_int32), ("key_id", C.c_int32 * K), ("traj_id", C.c_int32 * K), ("n_rows", C.c_int32 * K),
                ("cut_index_pos", C.c_int32), ("cut_layer", C.c_int32), ("vel_plan", C.c_double), ("acc_plan", C.c_double),
                ("n_vel_course", C.c_int32), ("n_ids", C.c_int32), ("id_key", C.c_int32 * K), ("id_val", C.c_int32 * K),
                ("traj", _pf64 * K), ("vel_course", _pf64)]


# ltpl_config_online.ini of the reference (params/ltpl_config_online.ini) and the Graph_LTPL.graph_init defaults
DEFAULT_CONFIG = dict(w_last_edges=(0.0, 0.5, 0.8), v_max_offset=0.1, delaycomp=0.1, calc_time_safety=2.0,
                      calc_time_buffer_len=5, filt_window_width=1, dyn_model_exp=1.0, drag_coeff=0.85, m_veh=1000.0,
                      follow_control_type="PD", follow_control_params={"c_p": 1.25, "k_d": 0.025, "k_p": 0.2})


class Planner(object):
    """``n_scen`` planners behind one C handle. ``lib`` / ``prefix`` / ``create`` exist so that the build container's tests
    can bind the same class to the host-logic harness (oracle/planner_host.py); the product default is libltpl_hip.so."""

    def __init__(self, backend, n_scen=1, lib=None, prefix="ltpl_planner_", create=None, **config):
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config)
        self.n_scen = int(n_scen)
        self.backend = backend
        self.lib = lib if lib is not None else backend.lib
        self._prefix = prefix
        self._w_last = _f64(list(cfg["w_last_edges"]) if len(cfg["w_last_edges"]) else [0.0])
        c = PlannerConfig()
        c.n_scen, c.n_w_last = self.n_scen, len(cfg["w_last_edges"])
        c.w_last_edges = _p(self._w_last, _pf64)
        c.v_max_offset, c.delaycomp, c.calc_time_safety = cfg["v_max_offset"], cfg["delaycomp"], cfg["calc_time_safety"]
        c.calc_time_buffer_len, c.filt_window_width = int(cfg["calc_time_buffer_len"]), int(cfg["filt_window_width"])
        c.dyn_model_exp, c.drag_coeff, c.m_veh = cfg["dyn_model_exp"], cfg["drag_coeff"], cfg["m_veh"]
        if cfg["follow_control_type"] not in ("PD", "PDtan"):
            raise ValueError('Unsupported control type "' + str(cfg["follow_control_type"]) + '"!')
        c.follow_control_type = 0 if cfg["follow_control_type"] == "PD" else 1
        fp = cfg["follow_control_params"]
        c.c_p, c.k_p, c.k_d, c.tan_w = fp["c_p"], fp["k_p"], fp["k_d"], fp.get("tan_w", 1.0)
        self.handle = C.c_void_p()
        self._declare()
        if create is not None:
            rc = create(C.byref(c), C.byref(self.handle))
        else:
            fn = self._fn("create")
            fn.argtypes = [C.c_void_p, C.POINTER(PlannerConfig), C.POINTER(C.c_void_p)]
            rc = fn(backend.handle, C.byref(c), C.byref(self.handle))
        if rc != 0:
            msg = self._fn("last_error")(None)
            raise _capi.BackendError("ltpl_planner_create failed (%s): %s" % (_capi._STATUS.get(rc, rc), (msg or b"").decode()))
        caps = PlannerCaps()
        self._check(self._fn("get_caps")(self.handle, C.byref(caps)))
        self.cap_rows, self.cap_nodes = int(caps.cap_rows), int(caps.cap_nodes)
        # query buffers, reused by every accessor call
        self._pp = [np.zeros((self.cap_rows, 5)) for _ in range(K)]
        self._co = [np.zeros((self.cap_nodes, 8)) for _ in range(K)]
        self._nd = [np.zeros((self.cap_nodes, 2), np.int32) for _ in range(K)]
        self._ni = [np.zeros(self.cap_nodes, np.int32) for _ in range(K)]
        self._tr = [np.zeros((self.cap_rows, 7)) for _ in range(K)]
        self._vc = np.zeros(self.cap_rows)
        self._pv, self._tv = PathsView(), TrajView()
        for k in range(K):
            self._pv.path_param[k], self._pv.coeff[k] = _p(self._pp[k], _pf64), _p(self._co[k], _pf64)
            self._pv.nodes[k], self._pv.node_idx[k] = _p(self._nd[k], _pi32), _p(self._ni[k], _pi32)
            self._tv.traj[k] = _p(self._tr[k], _pf64)
        self._tv.vel_course = _p(self._vc, _pf64)
        # per-tick input staging (see _Staging) and the input structs that point into it
        self._stage_paths, self._stage_vel = _Staging(), _Staging()
        self._pin, self._vin = PlannerPathsIn(), PlannerVelIn()

    def _fn(self, name):
        return getattr(self.lib, self._prefix + name)

    def _declare(self):
        f = self._fn
        f("destroy").argtypes = [C.c_void_p]
        f("get_caps").argtypes = [C.c_void_p, C.POINTER(PlannerCaps)]
        f("last_error").argtypes = [C.c_void_p]
        f("last_error").restype = C.c_char_p
        f("set_start").argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   _pi32, _pi32]
        f("calc_paths").argtypes = [C.c_void_p, C.POINTER(PlannerPathsIn)]
        f("calc_paths_begin").argtypes = [C.c_void_p, C.POINTER(PlannerPathsIn)]
        f("calc_paths_finish").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f("get_ref_idx").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f("calc_vel_profile").argtypes = [C.c_void_p, C.POINTER(PlannerVelIn)]
        f("get_paths").argtypes = [C.c_void_p, C.c_int32, C.POINTER(PathsView)]
        f("get_trajectories").argtypes = [C.c_void_p, C.c_int32, C.POINTER(TrajView)]

    def _check(self, rc):
        if rc != 0:
            msg = self._fn("last_error")(self.handle)
            raise _capi.BackendError("ltpl_planner: %s: %s" % (_capi._STATUS.get(rc, rc), (msg or b"").decode()))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self._fn("destroy")(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- OnlineTrajectoryHandler.set_initial_pose ---------------------------------------------------------------------
    def set_start(self, scen, pos, heading, vel=0.0, max_heading_offset=math.pi / 4):
        it, ch = C.c_int32(1), C.c_int32(1)
        self._check(self._fn("set_start")(self.handle, int(scen), float(pos[0]), float(pos[1]), float(heading), float(vel),
                                          float(max_heading_offset), C.byref(it), C.byref(ch)))
        return bool(it.value), bool(ch.value)

    # ---- Graph_LTPL.calc_paths --------------------------------------------------------------------------------------------
    def _pack_paths_in(self, prev_actions, t_now, vehicles, zone_gids):
        # plain lists, then one pack per staging array (no ndarray, no per-column address lookup: see _Staging)
        n = self.n_scen
        acts = [KEY_IDS.get(a, _capi.ACT_NONE) if isinstance(a, str) else _capi.ACT_NONE for a in prev_actions]
        ts = [float(t_now)] * n if isinstance(t_now, (int, float)) else [float(x) for x in np.asarray(t_now, dtype=np.float64).reshape(-1)]
        if len(acts) != n or len(ts) != n:
            raise ValueError("prev_actions / t_now: one entry per planner expected")
        rad, vel, px, py, pos_off, veh_off = [], [], [], [], [0], [0]
        for s in range(n):
            for r, v, pos in vehicles[s]:
                rad.append(float(r))
                vel.append(float(v))
                for x, y in (pos.tolist() if isinstance(pos, np.ndarray) else pos):
                    px.append(float(x))
                    py.append(float(y))
                pos_off.append(len(px))
            veh_off.append(len(rad))
        zone_off, zone = [0], []
        for s in range(n):
            if zone_gids is not None:
                z = zone_gids[s]
                zone.extend(z.tolist() if isinstance(z, np.ndarray) else [int(g) for g in z])
            zone_off.append(len(zone))
        if not rad:
            rad, vel = [0.0], [0.0]                            # (never an empty array behind a pointer)
        if not px:
            px, py = [0.0], [0.0]
        if not zone:
            zone = [0]
        st, i = self._stage_paths, self._pin
        st.put(ts + rad + vel + px + py, acts + veh_off + pos_off + zone_off + zone)
        nv, npos = len(rad), len(px)
        o = st.ad
        i.t_now = o; o += 8 * n
        i.veh_radius = o; o += 8 * nv
        i.veh_vel = o; o += 8 * nv
        i.pos_x = o; o += 8 * npos
        i.pos_y = o
        o = st.ai
        i.prev_action = o; o += 4 * n
        i.veh_off = o; o += 4 * len(veh_off)
        i.pos_off = o; o += 4 * len(pos_off)
        i.zone_off = o; o += 4 * len(zone_off)
        i.zone_gid = o
        return i, st

    def _pack_zones(self, zone_gids):
        zone_off, zone = [0], []
        for s in range(self.n_scen):
            if zone_gids is not None:
                z = zone_gids[s]
                zone.extend(z.tolist() if isinstance(z, np.ndarray) else z)
            zone_off.append(len(zone))
        return np.array(zone_off, np.int32), np.array(zone or [0], np.int32)

    def calc_paths(self, prev_actions, t_now, vehicles, zone_gids=None):
        """``prev_actions``: action name per planner; ``vehicles``: per planner a list of (radius, vel, positions (k, 2) with the
        own position first); ``zone_gids``: per planner the global node ids removed by the "overtaking_zones" filter."""
        i, keep = self._pack_paths_in(prev_actions, t_now, vehicles, zone_gids)
        self._check(self._fn("calc_paths")(self.handle, C.byref(i)))

    def calc_paths_begin(self, prev_actions, t_now, vehicles):
        """First half of calc_paths: objects in, start nodes determined (read them with ``start_nodes()``)."""
        i, keep = self._pack_paths_in(prev_actions, t_now, vehicles, None)
        self._check(self._fn("calc_paths_begin")(self.handle, C.byref(i)))

    def calc_paths_finish(self, zone_gids=None):
        zone_off, zone = self._pack_zones(zone_gids)
        self._check(self._fn("calc_paths_finish")(self.handle, zone_off.ctypes.data, zone.ctypes.data))

    def _path_keys(self, scen):
        v = PathsView()                                    # counts only: no buffers attached
        self._check(self._fn("get_paths")(self.handle, int(scen), C.byref(v)))
        return [KEY_NAMES[v.key_id[k]] for k in range(v.n_keys)]

    def start_node(self, scen=0):
        v = PathsView()                                    # counts only: no buffers attached
        self._check(self._fn("get_paths")(self.handle, int(scen), C.byref(v)))
        return [int(v.start_node[0]), int(v.start_node[1])]

    def get_ref_idx(self, pos_est, scen=0):
        """OnlineTrajectoryHandler.get_ref_idx (OTH.py:518-601) for all planners; returns planner ``scen``'s 5-tuple."""
        pos = np.asarray(pos_est, dtype=np.float64).reshape(self.n_scen, 2)
        px, py = _f64(pos[:, 0]), _f64(pos[:, 1])
        self._check(self._fn("get_ref_idx")(self.handle, px.ctypes.data, py.ctypes.data))
        ref = self.trajectories(scen)[2]
        return ref["cut_index_pos"], ref["cut_layer"], ref["vel_plan"], ref["vel_course"], ref["acc_plan"]

    # ---- Graph_LTPL.calc_vel_profile --------------------------------------------------------------------------------------
    def _pack_vel_in(self, pos_est, vel_est, vel_max=100.0, gg_scale=1.0, local_gg=(5.0, 5.0),
                     ax_max_machines=((100.0, 5.0),), safety_d=30.0, incl_emerg_traj=False):
        """The input struct of calc_vel_profile (ltpl_planner_vel_in) filled from the reference's arguments; returns (struct, objects
        that must stay alive until the C call returned)."""
        n = self.n_scen
        per = list(local_gg) if isinstance(local_gg, list) else [local_gg] * n
        if len(per) != n:
            raise ValueError("local_gg: one entry per planner expected")
        gg_off, gg_rows, const = None, None, []
        if any(isinstance(g, dict) for g in per):
            gg_off, chunks = [0], []
            for s_, g in enumerate(per):
                keys = self._path_keys(s_) if isinstance(g, dict) else []
                for k in range(K):
                    if k < len(keys) and keys[k] in g:
                        a = np.asarray(g[keys[k]][0] if isinstance(g[keys[k]], (list, tuple)) else g[keys[k]], dtype=np.float64)
                        if a.ndim != 2 or a.shape[1] != 2:
                            raise ValueError("local_gg['%s']: rows of (ax, ay) expected" % keys[k])
                        chunks.append(a)
                        gg_off.append(gg_off[-1] + a.shape[0])
                    else:
                        gg_off.append(gg_off[-1])
                const.append((5.0, 5.0) if isinstance(g, dict) else g)
            gg_rows = np.ascontiguousarray(np.concatenate(chunks + [np.zeros((1, 2))]))
            gg_off = np.array(gg_off, np.int32)
        else:
            const = per
        if any(type(g) not in (tuple, list) or len(g) != 2 for g in const):
            raise ValueError("Provided local_gg does not satisfy requested format! Read parameter documentation.")
        local_gg = ([float(g[0]) for g in const], [float(g[1]) for g in const])
        flat = np.asarray(pos_est, dtype=np.float64).reshape(-1).tolist()
        if len(flat) != 2 * n:
            raise ValueError("pos_est: one (x, y) per planner expected")

        def bc(v):
            if isinstance(v, (int, float)):
                return [float(v)] * n
            a = np.asarray(v, dtype=np.float64).reshape(-1).tolist()
            return a * n if len(a) == 1 else a
        cols = [flat[0::2], flat[1::2], bc(vel_est), bc(vel_max), bc(gg_scale), bc(local_gg[0]), bc(local_gg[1]), bc(safety_d)]
        if any(len(c) != n for c in cols):
            raise ValueError("calc_vel_profile: scalars or one value per planner expected")
        axm = np.asarray(ax_max_machines, dtype=np.float64).reshape(-1).tolist()
        if len(axm) < 2 or len(axm) % 2:
            raise ValueError("ax_max_machines: rows [v, ax] expected")
        emerg = [int(bool(incl_emerg_traj))] * n if isinstance(incl_emerg_traj, (bool, int)) else [int(bool(e)) for e in incl_emerg_traj]
        if len(emerg) != n:
            raise ValueError("incl_emerg_traj: a bool or one value per planner expected")
        st, i = self._stage_vel, self._vin
        dbl = []
        for c in cols:
            dbl += c
        st.put(dbl + axm, emerg)
        o = st.ad
        i.pos_est_x = o; i.pos_est_y = o + 8 * n; i.vel_est = o + 16 * n; i.vel_max = o + 24 * n
        i.gg_scale = o + 32 * n; i.gg_ax = o + 40 * n; i.gg_ay = o + 48 * n; i.safety_d = o + 56 * n
        i.ax_max_machines = o + 64 * n
        i.n_ax_max_machines = len(axm) // 2
        i.incl_emerg_traj = st.ai
        i.gg_row_off = None if gg_off is None else gg_off.ctypes.data
        i.gg_rows = None if gg_rows is None else gg_rows.ctypes.data
        return i, (st, gg_off, gg_rows)

    def calc_vel_profile(self, pos_est, vel_est, vel_max=100.0, gg_scale=1.0, local_gg=(5.0, 5.0),
                         ax_max_machines=((100.0, 5.0),), safety_d=30.0, incl_emerg_traj=False):
        """``local_gg``: the constant form (ax, ay) -- or, location dependent friction (OTH.py:649-666), a dict {action id: [ndarray
        (rows, 2)]} with one row [ax, ay] per coordinate of that action's path (``paths(scen)['path_param'][key]``); for a batch a
        list with one such dict (or tuple) per planner. ``incl_emerg_traj``: bool or one per planner."""
        i, keep = self._pack_vel_in(pos_est, vel_est, vel_max=vel_max, gg_scale=gg_scale, local_gg=local_gg,
                                    ax_max_machines=ax_max_machines, safety_d=safety_d, incl_emerg_traj=incl_emerg_traj)
        self._check(self._fn("calc_vel_profile")(self.handle, C.byref(i)))

    # ---- accessors ----------------------------------------------------------------------------------------------------------
    def paths(self, scen=0):
        """State after calc_paths of planner ``scen``: dict with the reference's structures (OTH.py:509-516)."""
        v = self._pv
        self._check(self._fn("get_paths")(self.handle, int(scen), C.byref(v)))
        nk = v.n_keys
        key_id, n_rows, n_nodes, red = v.key_id[:nk], v.n_rows[:nk], v.n_nodes[:nk], v.red_len[:nk]     # ctypes arrays -> lists, once
        coi = v.closest_obj_index
        out = {"keys": [], "path_param": {}, "coeff": {}, "nodes": {}, "node_idx": {}, "red_len": {},
               "start_node": v.start_node[:2], "const_rows": v.const_rows, "closest_obj_index": None if coi < 0 else coi}
        for k in range(nk):
            name = KEY_NAMES[key_id[k]]
            nr, nn = n_rows[k], n_nodes[k]
            out["keys"].append(name)
            out["path_param"][name] = self._pp[k][:nr].copy()
            out["coeff"][name] = self._co[k][:max(nn - 1, 0)].copy()
            nodes = self._nd[k][:nn].tolist()
            if nn and nodes[0][0] < 0:                          # the start spline's pseudo node [None, None] (OTH.py:267)
                nodes = [[None if a < 0 else a, None if b < 0 else b] for a, b in nodes]
            out["nodes"][name] = nodes
            out["node_idx"][name] = self._ni[k][:nn].tolist()
            out["red_len"][name] = bool(red[k])
        return out

    def trajectories(self, scen=0):
        """(action_set, action_set_id, ref_idx) of planner ``scen`` after calc_vel_profile (OTH.py:1040, :601)."""
        v = self._tv
        self._check(self._fn("get_trajectories")(self.handle, int(scen), C.byref(v)))
        action_set, ids = {}, {}
        nk, ni = v.n_keys, v.n_ids
        key_id, n_rows = v.key_id[:nk], v.n_rows[:nk]
        for k in range(nk):
            action_set[KEY_NAMES[key_id[k]]] = [self._tr[k][:n_rows[k]].copy()]
        id_key, id_val = v.id_key[:ni], v.id_val[:ni]
        for k in range(ni):
            ids[KEY_NAMES[id_key[k]]] = id_val[k]
        ref = {"cut_index_pos": v.cut_index_pos, "cut_layer": v.cut_layer, "vel_plan": v.vel_plan,
               "acc_plan": v.acc_plan, "vel_course": self._vc[:v.n_vel_course].copy()}
        return action_set, ids, ref
