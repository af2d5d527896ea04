"""
Host-side mirror of seam (1): ``main_online_path_gen`` (graph_ltpl/online_graph/src/main_online_path_gen.py:11-334)
with the SAME signature, argument meaning and return structure, backed by the HIP library.

What stays on the host (cheap, O(#objects), Python objects in / out):
  * zone bookkeeping of gen_local_node_template.py:42-99 (which nodes the "overtaking_zones" filter removes; the list is
    only rebuilt when a zone is new / disabled or when there are no zones, exactly like the reference)
  * the constant-path-segment test of main_online_path_gen.py:76-122 (three get_s_coord projections per object)
  * unpacking of the flat device buffers into the dict-of-lists structures OnlineTrajectoryHandler consumes
Everything else -- obstacle/edge masking, filters, action-template choice, graph search with horizon back-off, spline
gather / re-solve / heading+curvature -- runs in one launch of the path kernel (csrc/ltpl_hip.hip).
"""
import logging
import numpy as np

from . import _capi
from .geometry import get_s_coord
from .lattice import Lattice

UNBLOCK_N_LAYERS_WHEN_IN_ZONE = 4          # gen_local_node_template.py:9
BLOCK_N_LAYERS_WHEN_REMOVING_ZONE = 0      # gen_local_node_template.py:10


class OnlinePathGenerator(object):
    """Callable drop-in for ``main_online_path_gen``; one instance per planner (it owns the zone-filter memory)."""

    def __init__(self, lattice: Lattice, backend):
        self.lat = lattice
        self.backend = backend
        self._zone_gids = np.zeros(0, dtype=np.int32)      # the reference's persistent "overtaking_zones" filter
        self._result = None
        self.last_result = None
        self.last_batch = None

    # -- gen_local_node_template.py:42-99 --------------------------------------------------------------------------
    def _update_zones(self, graph_base, start_node, obj_zone):
        if not (not obj_zone or not all([zone.processed for zone in obj_zone])
                or any([zone.disabled for zone in obj_zone])):
            return
        num_layers = self.lat.num_layers
        layer_ids_total, node_ids_total = [], []
        for zone in obj_zone:
            layer_ids, node_ids = zone.get_blocked_nodes(graph_base=graph_base)
            if not zone.processed or zone.disabled:
                n = UNBLOCK_N_LAYERS_WHEN_IN_ZONE if not zone.processed else BLOCK_N_LAYERS_WHEN_REMOVING_ZONE
                la = np.array(layer_ids)
                if (start_node[0] + n) <= num_layers:
                    u_l = np.logical_and(la >= start_node[0], la < (start_node[0] + n))
                else:
                    u_l = np.logical_or(np.logical_and(la >= start_node[0], la < num_layers),
                                        np.logical_and(la >= 0, la < ((start_node[0] + n) % (num_layers - 1) - 1)))
                if not zone.processed:
                    if any(u_l) and not zone.fixed:
                        logging.getLogger("local_trajectory_logger").\
                            critical("Vehicle within provided zone, unblock active!")
                        layer_ids = list(np.array(layer_ids)[~np.array(u_l)])
                        node_ids = list(np.array(node_ids)[~np.array(u_l)])
                    zone.set_processed()
                if zone.disabled:
                    if any(u_l):
                        layer_ids = list(np.array(layer_ids)[np.array(u_l)])
                        node_ids = list(np.array(node_ids)[np.array(u_l)])
                    else:
                        layer_ids = []
                        node_ids = []
                    zone.update_blocked_nodes(layer_ids=layer_ids, node_ids=node_ids)
                    zone.update_bound_coords(bound_l_coord=[0.0, 0.0], bound_r_coord=[0.0, 0.0])
            layer_ids_total.extend(layer_ids)
            node_ids_total.extend(node_ids)
        self.set_zone_nodes(layer_ids_total, node_ids_total)

    def set_zone_nodes(self, layer_ids, node_ids):
        """Replace the removed-node set of the "overtaking_zones" filter (GraphBase.remove_nodes_filter, 713-745)."""
        lat = self.lat
        gids = []
        for l, n in zip(layer_ids, node_ids):
            l, n = int(l), int(n)
            if 0 <= l < lat.num_layers and 0 <= n < lat.nodes_in_layer[l]:   # unknown names are simply not matched
                gids.append(int(lat.layer_off[l]) + n)
        self._zone_gids = np.array(sorted(set(gids)), dtype=np.int32)

    # -- main_online_path_gen.py:76-122 ----------------------------------------------------------------------------
    def _const_segment_test(self, obj_veh, const_path_seg, pos_est):
        obj_in_const_path = False
        object_besides_const_path = False
        const_closest = None
        lat = self.lat
        if const_path_seg is not None and np.size(const_path_seg, axis=0) >= 2:
            pos_start = pos_est if pos_est is not None else const_path_seg[0, 0:2]
            s_start, _ = get_s_coord(ref_line=lat.raceline, pos=pos_start, s_array=lat.s_raceline, closed=True)
            s_end, _ = get_s_coord(ref_line=lat.raceline, pos=const_path_seg[-1, 0:2], s_array=lat.s_raceline,
                                   closed=True)
            smallest_obj_dist = np.inf
            for obj_idx, vehicle in enumerate(obj_veh):
                s_obj, _ = get_s_coord(ref_line=lat.raceline, pos=vehicle.get_pos(), s_array=lat.s_raceline,
                                       closed=True)
                if s_start <= s_obj <= s_end or (s_start > s_end and (s_obj > s_start or s_obj < s_end)):
                    object_besides_const_path = True
                    if s_obj < s_start:
                        obj_dist = s_obj + lat.s_raceline[-1] - s_start
                    else:
                        obj_dist = s_obj - s_start
                    # the reference overwrites its closest_obj_index with the first besides-object unconditionally
                    # (smallest_obj_dist starts at inf, :96,113), so the result does not depend on the graph stage
                    if const_closest is None or obj_dist < smallest_obj_dist:
                        const_closest = obj_idx
                        smallest_obj_dist = obj_dist
                    obstacle_ref = np.power(vehicle.get_radius() + lat.veh_width / 2, 2)
                    distances2 = np.power(const_path_seg[:, 0] - vehicle.get_pos()[0], 2) + np.power(
                        const_path_seg[:, 1] - vehicle.get_pos()[1], 2)
                    if any(distances2 <= obstacle_ref):
                        obj_in_const_path = True
        return obj_in_const_path, object_besides_const_path, const_closest

    def scenario(self, start_node, obj_veh, action_sets=True, last_action_id=None, const_path_seg=None,
                 pos_est=None, last_solution_nodes=None):
        """Pack one call into the scenario dict understood by ``_capi.PathsBatch``."""
        in_const, besides, const_closest = self._const_segment_test(obj_veh, const_path_seg, pos_est)
        vehicles = []
        for vehicle in obj_veh:
            pos = np.asarray(vehicle.get_pos(), dtype=np.float64).reshape(1, 2)
            pred = vehicle.get_prediction()
            pred = np.zeros((0, 2)) if pred is None else np.asarray(pred, dtype=np.float64).reshape(-1, 2)
            vehicles.append((float(vehicle.get_radius()), np.vstack((pos, pred))))
        return {"start_node": (int(start_node[0]), int(start_node[1])), "action_sets": bool(action_sets),
                "obj_in_const": in_const, "obj_besides": besides, "last_action": last_action_id,
                "const_closest": const_closest,
                "psi_s": None if const_path_seg is None else float(const_path_seg[-1, 2]),
                "vehicles": vehicles, "zone_gids": self._zone_gids, "last_nodes": last_solution_nodes}

    def __call__(self, graph_base, start_node, obj_veh, obj_zone, action_sets=True, last_action_id=None,
                 max_solutions=1, const_path_seg=None, pos_est=None, last_solution_nodes=None, w_last_edges=()):
        if max_solutions != 1:
            # GraphBase.search_graph returns one path per target anyway (GraphBase.py:831)
            logging.getLogger("local_trajectory_logger").debug("max_solutions > 1 has no effect")
        self._update_zones(graph_base, start_node, obj_zone)
        sc = self.scenario(start_node, obj_veh, action_sets, last_action_id, const_path_seg, pos_est,
                           last_solution_nodes)
        batch = _capi.PathsBatch([sc], w_last_edges=w_last_edges)
        if self._result is None:
            self._result = self.backend.new_paths_result(1)
        res = self.backend.plan_paths(batch, self._result)
        self.last_result, self.last_batch = res, batch
        out = res.action_sets(0, int(start_node[0]), self.lat.num_layers)
        log = logging.getLogger("local_trajectory_logger")
        for a in range(int(res.n_actions[0])):
            name = _capi.ACTION_NAMES.get(int(res.action_id[0, a]), "?")
            if res.reduced[0, a]:
                log.info("No feasible solution for '" + name + "'! Reduced planning horizon!")
            if not res.valid[0, a]:
                log.debug("Action set '" + name + "' is empty! No path solution was found.")
        return out
