"""
Host-side mirror of seam (1): ``main_online_path_gen`` (graph_ltpl/online_graph/src/main_online_path_gen.py:11-334)
with the SAME signature, argument meaning and return structure, backed by the HIP library.

What stays in Python (Python objects in / out):
  * zone bookkeeping of gen_local_node_template.py:42-99 on the reference's own ZoneObject instances (zones.py)
  * unpacking of the flat device buffers into the dict-of-lists structures OnlineTrajectoryHandler consumes
The constant-path-segment test of main_online_path_gen.py:76-122 (projections of ego / segment end / objects on the race line)
runs behind the C ABI (ltpl_const_segment_test).
Everything else -- obstacle/edge masking, filters, action-template choice, graph search with horizon back-off, spline
gather / re-solve / heading+curvature -- runs in one launch of the path kernel (csrc/ltpl_hip.hip).
"""
import logging
import numpy as np

from . import _capi
from .lattice import Lattice
from .zones import ZoneFilter


class OnlinePathGenerator(object):
    """Callable drop-in for ``main_online_path_gen``; one instance per planner (it owns the zone-filter memory)."""

    def __init__(self, lattice: Lattice, backend):
        self.lat = lattice
        self.backend = backend
        self.zones = ZoneFilter(lattice)                   # the reference's persistent "overtaking_zones" filter
        self._result = None
        self.last_result = None
        self.last_batch = None

    def set_zone_nodes(self, layer_ids, node_ids):
        self.zones.set_nodes(layer_ids, node_ids)

    def scenario(self, start_node, obj_veh, action_sets=True, last_action_id=None, const_path_seg=None,
                 pos_est=None, last_solution_nodes=None):
        """Pack one call into the scenario dict understood by ``_capi.PathsBatch``. The constant-segment test
        (main_online_path_gen.py:76-122) runs behind the C ABI (ltpl_const_segment_test)."""
        vehicles = []
        for vehicle in obj_veh:
            pos = np.asarray(vehicle.get_pos(), dtype=np.float64).reshape(1, 2)
            pred = vehicle.get_prediction()
            pred = np.zeros((0, 2)) if pred is None else np.asarray(pred, dtype=np.float64).reshape(-1, 2)
            vehicles.append((float(vehicle.get_radius()), np.vstack((pos, pred))))
        in_const, besides, const_closest = self.backend.const_segment_test(const_path_seg, pos_est, vehicles)
        return {"start_node": (int(start_node[0]), int(start_node[1])), "action_sets": bool(action_sets),
                "obj_in_const": in_const, "obj_besides": besides, "last_action": last_action_id,
                "const_closest": const_closest,
                "psi_s": None if const_path_seg is None else float(const_path_seg[-1, 2]),
                "vehicles": vehicles, "zone_gids": self.zones.gids, "last_nodes": last_solution_nodes}

    def __call__(self, graph_base, start_node, obj_veh, obj_zone, action_sets=True, last_action_id=None,
                 max_solutions=1, const_path_seg=None, pos_est=None, last_solution_nodes=None, w_last_edges=()):
        if max_solutions != 1:
            # GraphBase.search_graph returns one path per target anyway (GraphBase.py:831)
            logging.getLogger("local_trajectory_logger").debug("max_solutions > 1 has no effect")
        self.zones.refresh(graph_base, int(start_node[0]), obj_zone)
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
