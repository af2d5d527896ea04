"""
Tick log in the reference's on-disk format: one header + one ';'-separated row of JSON cells per
planning tick, column for column what ``Logging.log_onlinegraph`` writes
(graph_ltpl/helper_funcs/src/Logging.py:35-37,47-126), so that the stock log viewer
(graph_ltpl/visualization/src/visualize_graph_log.py:66-130) opens logs produced through the planner entry points, and logs
written by the reference itself can be read back here.

  TickLogWriter   writes rows (from a ``Planner`` or from explicit values)
  read_log        parses a log (ours or the reference's) into dict rows
  revalidate      re-runs seam (1) for EVERY logged tick in one batched launch and compares the node lists with the logged
                  ones -- the batched form of the viewer's RECALC_VALIDATION (visualize_graph_log.py:210-234); logs double as
                  regression vectors
"""
import json
import numpy as np

from . import _capi

COLUMNS = ("time", "s_coord", "start_node", "obj_veh", "obj_zone", "nodes_list", "s_list", "pos_list", "vel_list", "a_list",
           "psi_list", "kappa_list", "traj_id", "clip_pos", "action_id_prev", "traj_id_prev", "const_path_seg")


def _jsonable(obj):
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    raise TypeError('Not serializable (type: ' + str(type(obj)) + ')')


class TickLogWriter(object):
    def __init__(self, log_path: str, graph_id: str = "ltpl-hip"):
        self.path = log_path
        self._zone_sig, self._zone_stamp = None, None
        with open(log_path, "w+") as fh:
            fh.write("#" + str(graph_id) + "\n" + ";".join(COLUMNS))

    def write(self, time, s_coord, start_node, obj_veh, obj_zone, nodes_list, traj_set, traj_id, clip_pos, action_id_prev,
              traj_id_prev=0, const_path_seg=None):
        """``obj_veh``: rows [id, pos, psi, radius, vel, prediction]; ``obj_zone``: rows [[layers, nodes], [bound_l, bound_r]];
        ``traj_set``: {key: [ndarray (n, 7) = s, x, y, psi, kappa, vx, ax]} as returned by calc_vel_profile."""
        sig = json.dumps(obj_zone, default=_jsonable)
        if sig != self._zone_sig:
            zone_cell, self._zone_sig, self._zone_stamp = obj_zone, sig, str(time)
        else:
            zone_cell = ["no update since", self._zone_stamp]         # Logging.py:88-97

        def col(c):
            return {k: [np.asarray(t)[:, c] for t in v] for k, v in traj_set.items()}
        pos = {k: [np.asarray(t)[:, 1:3] for t in v] for k, v in traj_set.items()}
        if const_path_seg is not None:
            const_path_seg = np.asarray(const_path_seg)[:, 0:2]
        cells = [str(time), str(s_coord)] + [json.dumps(c, default=_jsonable) for c in (
            start_node, obj_veh, zone_cell, nodes_list, col(0), pos, col(5), col(6), col(3), col(4), traj_id, list(clip_pos),
            action_id_prev, traj_id_prev, const_path_seg)]
        with open(self.path, "a") as fh:
            fh.write("\n" + ";".join(cells))

    @staticmethod
    def snapshot_paths(planner, scen=0):
        """What the row needs from ``calc_paths``: call this BETWEEN calc_paths and calc_vel_profile. The reference logs the start node,
        the node lists and the constant path segment as calc_paths returned them (Graph_LTPL.py:336-340, :445-447); the velocity stage
        trims the planner's stored paths and node lists by the cut layer afterwards (OTH.py:714-731)."""
        p = planner.paths(scen)
        const = None
        if p["const_rows"] >= 0 and p["keys"]:
            const = p["path_param"][p["keys"][0]][:max(p["const_rows"], 0), :].copy()
        return {"start_node": list(p["start_node"]), "nodes": {k: [p["nodes"][k]] for k in p["keys"]}, "const_path_seg": const}

    def write_planner_tick(self, planner, backend, time, pos_est, action_id_prev, vehicles, zone_layers=(), zone_nodes=(),
                           scen=0, export_rows=None, paths_snapshot=None):
        """One row from the state of planner ``scen`` after calc_vel_profile. ``vehicles`` = [(radius, vel, positions)].
        ``paths_snapshot``: result of ``snapshot_paths`` taken right after calc_paths (recommended: without it the node lists and the
        constant segment are read from the state the velocity stage has already trimmed, which is only the same while cut_layer = 0)."""
        traj, ids, ref = planner.trajectories(scen)
        if paths_snapshot is not None:
            const = paths_snapshot["const_path_seg"]
            if const is not None:
                const = const[ref["cut_index_pos"]:, :]               # Graph_LTPL.py:445-447
            if export_rows is not None:
                traj = {k: [t[:export_rows] for t in v] for k, v in traj.items()}
            obj_veh = [[k, list(map(float, pos[0])), 0.0, float(r), float(v), np.asarray(pos[1:]).reshape(-1, 2)]
                       for k, (r, v, pos) in enumerate(vehicles)]
            zones = [[[list(map(int, zone_layers)), list(map(int, zone_nodes))], [[0.0, 0.0], [0.0, 0.0]]]] if len(zone_layers) else []
            self.write(time, backend.raceline_s(pos_est), paths_snapshot["start_node"], obj_veh, zones, paths_snapshot["nodes"],
                       traj, ids, list(map(float, pos_est)), action_id_prev, 0, const)
            return
        p = planner.paths(scen)
        if export_rows is not None:
            traj = {k: [t[:export_rows] for t in v] for k, v in traj.items()}
        obj_veh = [[k, list(map(float, pos[0])), 0.0, float(r), float(v), np.asarray(pos[1:]).reshape(-1, 2)]
                   for k, (r, v, pos) in enumerate(vehicles)]
        zones = [[[list(map(int, zone_layers)), list(map(int, zone_nodes))], [[0.0, 0.0], [0.0, 0.0]]]] if len(zone_layers) else []
        const = None
        if p["const_rows"] >= 0 and p["keys"]:
            const = p["path_param"][p["keys"][0]][ref["cut_index_pos"]:max(p["const_rows"] - 0, 0), :]
        self.write(time, backend.raceline_s(pos_est), p["start_node"], obj_veh, zones, {k: [p["nodes"][k]] for k in p["keys"]},
                   traj, ids, list(map(float, pos_est)), action_id_prev, 0, const)


def read_log(log_path: str):
    """(graph_id, rows): every row a dict column -> decoded JSON value (time / s_coord as float)."""
    with open(log_path) as fh:
        graph_id = fh.readline().rstrip("\n").lstrip("#")
        header = fh.readline().rstrip("\n").split(";")
        rows = []
        for line in fh:
            line = line.rstrip("\n")
            if not line:
                continue
            cells = dict(zip(header, line.split(";")))
            row = {}
            for k, v in cells.items():
                row[k] = float(v) if k in ("time", "s_coord") else json.loads(v)
            rows.append(row)
    # zones are only logged when they change (Logging.py:88-97): resolve the back references
    last = []
    for r in rows:
        z = r.get("obj_zone", [])
        if z and z[0] == "no update since":
            r["obj_zone"] = last
        else:
            last = z
    return graph_id, rows


def revalidate(backend, lattice, rows, w_last_edges=(), context=False):
    """Re-plan every logged tick (start node, objects, zones, previous action; no constant segment, like the stock viewer's
    re-run) as ONE batch through seam (1) and compare node lists. A logged list may carry nodes of the constant segment in
    front (visualize_graph_log.py:226-230), so the re-planned list has to match its tail. Returns the list of mismatches.

    ``context=True`` goes beyond the stock viewer: a row's PREDECESSOR in the log holds what seam (1) was additionally given on that
    tick -- the previous solution's nodes from the start node on (cost discount ``w_last_edges``, OTH.py:386-393), the position
    estimate (``clip_pos`` of the row before, OTH.py:585) -- and the row itself the constant segment (x, y from the current position
    on: decides the [follow, left / right] template of main_online_path_gen.py:76-122). With them the re-run reproduces the logged
    node lists tick for tick, which turns a log into a strict regression vector (tests/test_tick_log.py)."""
    scen = []
    prev = None
    for r in rows:
        vehicles = []
        for o in r["obj_veh"]:
            pos = np.asarray(o[1], dtype=float).reshape(1, 2)
            pred = np.asarray(o[5], dtype=float).reshape(-1, 2) if o[5] is not None and len(o[5]) else np.zeros((0, 2))
            vehicles.append((float(o[3]), np.vstack((pos, pred))))
        gids = []
        for z in r["obj_zone"]:
            for l, n in zip(z[0][0], z[0][1]):
                if 0 <= l < lattice.num_layers and 0 <= n < lattice.nodes_in_layer[l]:
                    gids.append(int(lattice.layer_off[l]) + int(n))
        sc = {"start_node": tuple(r["start_node"]), "action_sets": True, "vehicles": vehicles,
              "zone_gids": sorted(set(gids)), "last_action": r["action_id_prev"], "last_nodes": None,
              "obj_in_const": False, "obj_besides": False, "const_closest": None, "psi_s": None}
        if context and prev is not None:
            last = prev["nodes_list"].get(r["action_id_prev"])
            if last is not None:
                nl = [list(v) for v in last[0]]
                if list(r["start_node"]) in nl:
                    sc["last_nodes"] = nl[nl.index(list(r["start_node"])):]
            seg = r.get("const_path_seg")
            if seg is not None and len(seg) >= 2:
                seg5 = np.zeros((len(seg), 5))                       # rows [x, y, psi, kappa, el]: the test only reads x, y
                seg5[:, 0:2] = np.asarray(seg, dtype=float).reshape(-1, 2)
                ic, bs, cc = backend.const_segment_test(seg5, prev["clip_pos"], vehicles)
                sc.update(obj_in_const=ic, obj_besides=bs, const_closest=cc)
        scen.append(sc)
        prev = r
    if not scen:
        return []
    res = backend.plan_paths(_capi.PathsBatch(scen, w_last_edges=w_last_edges))
    bad = []
    for i, r in enumerate(rows):
        nodes = res.action_sets(i, int(r["start_node"][0]), lattice.num_layers)[0]
        for key, logged in r["nodes_list"].items():
            got = nodes.get(key)
            if got is None:
                continue                # the key came from a template that needs the constant segment (not in the log)
            tail = [list(v) for v in logged[0]][-len(got[0]):]
            if tail != got[0]:
                bad.append((i, key))
    return bad
