"""
Content of the reference's "overtaking_zones" node filter: which lattice nodes the blocked zones currently remove
(graph_ltpl/online_graph/src/gen_local_node_template.py:42-99). The filter only changes when a zone arrives, leaves or there
are no zones at all; between those events the reference keeps the last filter (GraphBase.remove_nodes_filter, GraphBase.py:713-745)
-- here the removed set is kept as a sorted array of global node ids and handed to the device with every call.

The zone objects themselves (``ZoneObject``: processed / disabled / fixed flags, node lists) are the reference's and are only
touched through their public methods.
"""
import logging
import numpy as np

LAYERS_FREED_WHEN_INSIDE_NEW_ZONE = 4     # UNBLOCK_N_LAYERS_WHEN_IN_ZONE       gen_local_node_template.py:9
LAYERS_KEPT_OF_REMOVED_ZONE = 0           # BLOCK_N_LAYERS_WHEN_REMOVING_ZONE   gen_local_node_template.py:10


def layers_ahead_mask(layer_ids: np.ndarray, start_layer: int, n_ahead: int, num_layers: int) -> np.ndarray:
    """True for the zone entries whose layer lies in the ``n_ahead`` layers from ``start_layer`` on; the wrapped branch keeps
    the reference's bounds ``(start + n) % (num_layers - 1) - 1`` (gen_local_node_template.py:57-65)."""
    end = start_layer + n_ahead
    if end <= num_layers:
        return (layer_ids >= start_layer) & (layer_ids < end)
    wrapped_end = end % (num_layers - 1) - 1
    return (layer_ids >= start_layer) | (layer_ids < wrapped_end)


class ZoneFilter(object):
    def __init__(self, lattice):
        self.lat = lattice
        self.gids = np.zeros(0, dtype=np.int32)

    def set_nodes(self, layer_ids, node_ids):
        """Replace the removed-node set; names that do not exist in the lattice are simply not matched (GraphBase.py:713-745)."""
        lat = self.lat
        la, no = np.asarray(layer_ids, dtype=np.int64).reshape(-1), np.asarray(node_ids, dtype=np.int64).reshape(-1)
        ok = (la >= 0) & (la < lat.num_layers)
        la, no = la[ok], no[ok]
        ok = (no >= 0) & (no < lat.nodes_in_layer[la])
        self.gids = np.unique(lat.layer_off[la[ok]].astype(np.int64) + no[ok]).astype(np.int32)

    def refresh(self, graph_base, start_layer: int, obj_zone) -> None:
        zones = list(obj_zone) if obj_zone else []
        if zones and all(z.processed for z in zones) and not any(z.disabled for z in zones):
            return                                              # nothing arrived or left: the last filter stays
        L = self.lat.num_layers
        all_layers, all_nodes = [], []
        for z in zones:
            layers, nodes = z.get_blocked_nodes(graph_base=graph_base)
            layers, nodes = np.asarray(layers, dtype=np.int64).reshape(-1), np.asarray(nodes, dtype=np.int64).reshape(-1)
            fresh, leaving = not z.processed, bool(z.disabled)
            if fresh or leaving:
                ahead = layers_ahead_mask(layers, int(start_layer),
                                          LAYERS_FREED_WHEN_INSIDE_NEW_ZONE if fresh else LAYERS_KEPT_OF_REMOVED_ZONE, L)
                if fresh:
                    if ahead.any() and not z.fixed:
                        logging.getLogger("local_trajectory_logger").critical("Vehicle within provided zone, unblock active!")
                        layers, nodes, ahead = layers[~ahead], nodes[~ahead], ahead[~ahead]
                    z.set_processed()
                if leaving:
                    layers, nodes = layers[ahead], nodes[ahead]
                    z.update_blocked_nodes(layer_ids=[int(v) for v in layers], node_ids=[int(v) for v in nodes])
                    z.update_bound_coords(bound_l_coord=[0.0, 0.0], bound_r_coord=[0.0, 0.0])
            all_layers.append(layers)
            all_nodes.append(nodes)
        if all_layers:
            self.set_nodes(np.concatenate(all_layers), np.concatenate(all_nodes))
        else:
            self.set_nodes([], [])
