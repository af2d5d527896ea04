"""
ctypes binding of the fleet entry points (``ltpl_fleet_*``, include/ltpl_hip.h ABI v5): the planner of ``planner.py`` -- the
iterative memory of the reference's ``OnlineTrajectoryHandler`` (graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:24-1040) -- for
MANY vehicles on one lattice with the state in device memory. Every stage the host planner runs per vehicle on the CPU runs as a kernel
with one wave64 per planner (csrc/fleet_core.hpp); arguments, views and accessors are those of ``Planner``.

The tape form pre-uploads the inputs of many ticks and advances all planners through them without host synchronisation
(``tape_append`` / ``tape_append_groups`` / ``tape_run``): the closed-loop, state-carrying throughput of the hot path.
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from .planner import KEY_IDS, Planner, PlannerPathsIn, PlannerVelIn


class Fleet(Planner):
    def __init__(self, backend, n_planners, **config):
        Planner.__init__(self, backend, n_scen=n_planners, prefix="ltpl_fleet_", **config)

    def _declare(self):
        Planner._declare(self)
        f = self._fn
        f("set_start_range").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        f("tape_clear").argtypes = [C.c_void_p]
        f("tape_append").argtypes = [C.c_void_p, C.POINTER(PlannerPathsIn), C.POINTER(PlannerVelIn)]
        f("tape_run").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
        if hasattr(self.lib, "ltpl_fleet_digest"):
            f("digest").argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]

    def set_start_range(self, first, past_last, pos, heading, vel=0.0, max_heading_offset=math.pi / 4):
        """``set_start`` with the same pose for the planners [first, past_last) in one call (ltpl_fleet_set_start_range)."""
        it, ch = C.c_int32(1), C.c_int32(1)
        self._check(self._fn("set_start_range")(self.handle, int(first), int(past_last), float(pos[0]), float(pos[1]), float(heading),
                                                float(vel), float(max_heading_offset), C.byref(it), C.byref(ch)))
        return bool(it.value), bool(ch.value)

    # ---- tape ---------------------------------------------------------------------------------------------------------------
    def tape_clear(self):
        self._check(self._fn("tape_clear")(self.handle))

    def tape_append(self, prev_actions, t_now, vehicles, zone_gids, pos_est, vel_est, **vel_kwargs):
        """Inputs of one tick, arguments as ``calc_paths`` followed by ``calc_vel_profile``."""
        pi, keep1 = self._pack_paths_in(prev_actions, t_now, vehicles, zone_gids)
        vi, keep2 = self._pack_vel_in(pos_est, vel_est, **vel_kwargs)
        self._check(self._fn("tape_append")(self.handle, C.byref(pi), C.byref(vi)))

    @staticmethod
    def _machine_tables(vi, tables, idx, keep):
        """Machine limits per planner (ABI v6): ``tables`` = list of (rows, 2) arrays [v, ax], ``idx`` = table of every planner. One table:
        the plain form (ax_max_machines / n_ax_max_machines); several: stacked back to back with row offsets and the per-planner index."""
        f64, i32 = np.float64, np.int32
        tabs = [np.ascontiguousarray(np.asarray(t, f64).reshape(-1, 2)) for t in tables]
        stacked = np.ascontiguousarray(np.concatenate(tabs))
        vi.ax_max_machines, vi.n_ax_max_machines = stacked.ctypes.data, stacked.shape[0]
        keep.append(stacked)
        if len(tabs) > 1:
            off = np.ascontiguousarray(np.concatenate(([0], np.cumsum([t.shape[0] for t in tabs]))).astype(i32))
            ti = np.ascontiguousarray(np.asarray(idx, i32).reshape(-1))
            vi.n_ax_tables, vi.ax_table_off, vi.ax_table_idx = len(tabs), off.ctypes.data, ti.ctypes.data
            keep += [off, ti]
        else:
            vi.n_ax_tables, vi.ax_table_off, vi.ax_table_idx = 0, None, None

    def pack_groups(self, groups, ax_max_machines=((100.0, 5.0),)):
        """Input structs of one tick for planners that come in GROUPS with identical inputs (vectorised: no Python loop over planners).
        ``groups``: list of (count, dict) in planner order; dict keys: prev_action (name), t_now, vehicles [(radius, vel, positions)],
        zone_gids, pos_est, vel_est, vel_max, gg_scale, local_gg (ax, ay), safety_d, incl_emerg_traj. Returns (paths struct, velocity struct,
        keep-alive): pass the structs to ``calc_paths_packed`` / ``calc_vel_profile_packed`` / ``tape_append_packed``."""
        if sum(c for c, _ in groups) != self.n_scen:
            raise ValueError("pack_groups: the group sizes must add up to the number of planners")
        i32, f64 = np.int32, np.float64
        acts, ts, veh_cnt, pos_cnt, rad, vel, px, py, zcnt, zg = [], [], [], [], [], [], [], [], [], []
        vcols = [[] for _ in range(8)]
        emerg = []
        for cnt, g in groups:
            a = g["prev_action"]
            acts.append(np.full(cnt, KEY_IDS.get(a, _capi.ACT_NONE) if isinstance(a, str) else _capi.ACT_NONE, i32))
            ts.append(np.full(cnt, float(g["t_now"]), f64))
            vs = g["vehicles"]
            veh_cnt.append(np.full(cnt, len(vs), np.int64))
            pc = np.array([len(v[2]) for v in vs], np.int64)
            pos_cnt.append(np.tile(pc, cnt))
            rad.append(np.tile(np.array([float(v[0]) for v in vs], f64), cnt))
            vel.append(np.tile(np.array([float(v[1]) for v in vs], f64), cnt))
            pp = np.concatenate([np.asarray(v[2], f64).reshape(-1, 2) for v in vs]) if vs else np.zeros((0, 2))
            px.append(np.tile(pp[:, 0], cnt)); py.append(np.tile(pp[:, 1], cnt))
            z = np.asarray(sorted(set(int(q) for q in (g.get("zone_gids") or []))), i32)
            zcnt.append(np.full(cnt, len(z), np.int64)); zg.append(np.tile(z, cnt))
            lg = g.get("local_gg", (5.0, 5.0))
            if type(lg) not in (tuple, list) or len(lg) != 2:
                raise ValueError("Provided local_gg does not satisfy requested format! Read parameter documentation.")
            vals = (g["pos_est"][0], g["pos_est"][1], g["vel_est"], g.get("vel_max", 100.0), g.get("gg_scale", 1.0), lg[0], lg[1],
                    g.get("safety_d", 30.0))
            for k in range(8):
                vcols[k].append(np.full(cnt, float(vals[k]), f64))
            emerg.append(np.full(cnt, int(bool(g.get("incl_emerg_traj", False))), i32))

        def cat(lst, dt, pad):
            a = np.concatenate(lst).astype(dt) if lst else np.zeros(0, dt)
            return np.ascontiguousarray(a if a.size else np.full(1, pad, dt))

        def csr(counts):
            c = np.concatenate(counts) if counts else np.zeros(0, np.int64)
            return np.ascontiguousarray(np.concatenate(([0], np.cumsum(c))).astype(i32))
        arrs = dict(prev_action=cat(acts, i32, -1), t_now=cat(ts, f64, 0.0), veh_off=csr(veh_cnt), pos_off=csr(pos_cnt),
                    veh_radius=cat(rad, f64, 0.0), veh_vel=cat(vel, f64, 0.0), pos_x=cat(px, f64, 0.0), pos_y=cat(py, f64, 0.0),
                    zone_off=csr(zcnt), zone_gid=cat(zg, i32, 0))
        pi = PlannerPathsIn()
        for k, a in arrs.items():
            setattr(pi, k, a.ctypes.data)
        v = [cat(c, f64, 0.0) for c in vcols]
        em = cat(emerg, i32, 0)
        vi = PlannerVelIn()
        for name, a in zip(("pos_est_x", "pos_est_y", "vel_est", "vel_max", "gg_scale", "gg_ax", "gg_ay", "safety_d"), v):
            setattr(vi, name, a.ctypes.data)
        vi.incl_emerg_traj = em.ctypes.data
        vi.gg_row_off, vi.gg_rows = None, None
        # machine limits: the call's table, or -- a fleet of different cars -- a table per group (key "ax_max_machines" of the group)
        tables, tab_idx, keep_t = [], [], []
        for cnt, g in groups:
            t = np.asarray(g.get("ax_max_machines", ax_max_machines), f64).reshape(-1, 2)
            k = next((i for i, u in enumerate(tables) if u.shape == t.shape and np.array_equal(u, t)), None)
            if k is None:
                tables.append(t); k = len(tables) - 1
            tab_idx.append(np.full(cnt, k, i32))
        Fleet._machine_tables(vi, tables, np.concatenate(tab_idx), keep_t)
        return pi, vi, (arrs, v, em, keep_t)

    def pack_arrays(self, prev_action, t_now, veh_off, pos_off, veh_radius, veh_vel, pos_x, pos_y, zone_off, zone_gid,
                    pos_est, vel_est, vel_max=100.0, gg_scale=1.0, local_gg=(5.0, 5.0), safety_d=30.0, incl_emerg_traj=False,
                    ax_max_machines=((100.0, 5.0),), ax_tables=None, ax_table_idx=None, gg_row_off=None, gg_rows=None):
        """Input structs of one tick from the caller's own arrays (a simulator that holds its vehicles as arrays): the layout of
        ``ltpl_planner_paths_in`` / ``ltpl_planner_vel_in`` (include/ltpl_hip.h) -- ``prev_action`` action ids (LTPL_ACT_*) per planner,
        CSR offsets ``veh_off`` [n + 1] / ``pos_off`` [n_veh + 1] / ``zone_off`` [n + 1], own position of a vehicle first. Scalars are
        broadcast; ``vel_max`` may be an array (one value per planner). Different cars: ``ax_tables`` = list of machine tables and
        ``ax_table_idx`` = the table of every planner (instead of the one table ``ax_max_machines``). Location dependent friction (local_gg as a
        dict, OTH.py:649-666): ``gg_rows`` (rows, 2) = [ax, ay] per path coordinate and ``gg_row_off`` [n * 4 + 1] = the rows of planner p's
        k-th path key are gg_row_off[4 p + k] .. gg_row_off[4 p + k + 1] (none: that key drives with the constant ``local_gg``). Returns (paths struct, velocity struct,
        keep-alive) like ``pack_groups``."""
        n, i32, f64 = self.n_scen, np.int32, np.float64

        def arr(a, dt, size=None, pad=None):
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dt), (size,)) if size is not None else np.asarray(a, dt).reshape(-1))
            return a if a.size else np.full(1, pad if pad is not None else 0, dt)
        pe = np.asarray(pos_est, f64).reshape(-1, 2)
        if pe.shape[0] == 1:
            pe = np.broadcast_to(pe, (n, 2))
        if pe.shape[0] != n:
            raise ValueError("pos_est: one (x, y) per planner expected")
        arrs = dict(prev_action=arr(prev_action, i32, n), t_now=arr(t_now, f64, n), veh_off=arr(veh_off, i32), pos_off=arr(pos_off, i32),
                    veh_radius=arr(veh_radius, f64), veh_vel=arr(veh_vel, f64), pos_x=arr(pos_x, f64), pos_y=arr(pos_y, f64),
                    zone_off=arr(zone_off, i32), zone_gid=arr(zone_gid, i32))
        if arrs["veh_off"].size != n + 1 or arrs["zone_off"].size != n + 1 or arrs["pos_off"].size != int(arrs["veh_off"][-1]) + 1:
            raise ValueError("pack_arrays: veh_off / zone_off need n + 1 entries, pos_off one more than the number of vehicles")
        pi = PlannerPathsIn()
        for k, a in arrs.items():
            setattr(pi, k, a.ctypes.data)
        if type(local_gg) not in (tuple, list) or len(local_gg) != 2:
            raise ValueError("Provided local_gg does not satisfy requested format! Read parameter documentation.")
        v = [np.ascontiguousarray(pe[:, 0]), np.ascontiguousarray(pe[:, 1]), arr(vel_est, f64, n), arr(vel_max, f64, n), arr(gg_scale, f64, n),
             arr(local_gg[0], f64, n), arr(local_gg[1], f64, n), arr(safety_d, f64, n)]
        em = arr(np.asarray(incl_emerg_traj).astype(i32), i32, n)
        vi = PlannerVelIn()
        for name, a in zip(("pos_est_x", "pos_est_y", "vel_est", "vel_max", "gg_scale", "gg_ax", "gg_ay", "safety_d"), v):
            setattr(vi, name, a.ctypes.data)
        vi.incl_emerg_traj = em.ctypes.data
        vi.gg_row_off, vi.gg_rows = None, None
        keep_t = []
        if gg_row_off is not None or gg_rows is not None:
            go = np.ascontiguousarray(np.asarray(gg_row_off, i32).reshape(-1))
            gr = np.ascontiguousarray(np.asarray(gg_rows, f64).reshape(-1, 2))
            if go.size != n * _capi.PLANNER_MAX_KEYS + 1 or gr.shape[0] < int(go[-1]):
                raise ValueError("pack_arrays: gg_row_off needs n * %d + 1 entries and gg_rows gg_row_off[-1] rows" % _capi.PLANNER_MAX_KEYS)
            if gr.shape[0] == 0:
                gr = np.zeros((1, 2))
            vi.gg_row_off, vi.gg_rows = go.ctypes.data, gr.ctypes.data
            keep_t += [go, gr]
        if ax_tables is not None:
            if ax_table_idx is None or len(np.asarray(ax_table_idx).reshape(-1)) != n:
                raise ValueError("pack_arrays: ax_table_idx needs one entry per planner")
            Fleet._machine_tables(vi, list(ax_tables), ax_table_idx, keep_t)
        else:
            Fleet._machine_tables(vi, [ax_max_machines], None, keep_t)
        return pi, vi, (arrs, v, em, keep_t)

    def calc_paths_packed(self, pi):
        """``calc_paths`` on an input struct of ``pack_groups`` (a caller that already holds its fleet's inputs as arrays pays no packing)."""
        self._check(self._fn("calc_paths")(self.handle, C.byref(pi)))

    def calc_vel_profile_packed(self, vi):
        self._check(self._fn("calc_vel_profile")(self.handle, C.byref(vi)))

    def tape_append_groups(self, groups, ax_max_machines=((100.0, 5.0),)):
        """Inputs of one tick for planners that come in groups with identical inputs (see ``pack_groups``)."""
        pi, vi, keep = self.pack_groups(groups, ax_max_machines)
        self._check(self._fn("tape_append")(self.handle, C.byref(pi), C.byref(vi)))

    def tape_append_packed(self, pi, vi):
        """Append the input structs of ``pack_groups`` / ``pack_arrays`` (both calls of one tick) to the tape."""
        self._check(self._fn("tape_append")(self.handle, C.byref(pi), C.byref(vi)))

    DIGEST_DOUBLES = 8 + 9 * _capi.PLANNER_MAX_KEYS

    def digest(self):
        """Digest of EVERY planner's last tick, computed on the device (ltpl_fleet_digest): array [n_planners, DIGEST_DOUBLES] --
        [0] error word, [1] cut_index_pos, [2] cut_layer, [3] n_keys, [4] n_ids, [5] vel_plan, [6] n_vel_course, [7] acc_plan, per key k:
        [8 + 7 k ..] key id, trajectory id, rows, s_end, vx[0], vx[-1], sum(vx); per id k: [8 + 7 K + 2 k ..] key id, id value.
        ``tick_replay.check_digests`` compares it with a tick of a recording for all planners at once."""
        out = np.zeros((self.n_scen, self.DIGEST_DOUBLES), np.float64)
        self._check(self._fn("digest")(self.handle, out.ctypes.data_as(C.POINTER(C.c_double)), int(self.DIGEST_DOUBLES)))
        return out

    def tape_run(self, first, count):
        """Advance all planners through ticks [first, first + count) of the tape; returns the device time in ms."""
        ms = C.c_float(0.0)
        self._check(self._fn("tape_run")(self.handle, int(first), int(count), C.byref(ms)))
        return float(ms.value)
