"""
Struct-of-arrays view of the offline lattice ("offline_graph") -- the thing that is uploaded once to HBM.

The reference keeps the lattice inside an igraph object wrapped by ``GraphBase`` (graph_ltpl/data_objects/GraphBase.py);
every vertex carries ``position, psi, raceline, node_id, layer_id`` (GraphBase.py:163-168), every edge
``spline_coeff, spline_length, spline_param, offline_cost`` (GraphBase.py:409-439) and every layer has one virtual goal
vertex whose edge cost is ``abs(raceline_index - node) * lat_resolution * virt_goal_node_cost`` (GraphBase.py:188-194).
Edges only connect layer i to layer i+1 (gen_edges.py:52-61), so the device layout is:

  * nodes:   global id ``layer_off[l] + n``; ``node_pos``, ``node_psi``, ``vgoal_cost`` (virtual vertices dropped)
  * edges:   CSC per destination node (in-edges of node v are ``in_ptr[v]:in_ptr[v+1]``, sorted by source node id), so
             all edges of one layer transition are contiguous, and so are their samples
  * samples: rows of the per-edge ``spline_param`` [x, y, psi, kappa, el_length] concatenated in edge order
  * layers:  ``raceline_index, s_raceline, refline, raceline, vel_raceline`` (GraphBase.py:93-119)
  * glob_rl: fine global race line [s, x, y, kappa, vel] used by the follow-mode profile (GraphBase.py:111)

``Lattice.from_graph_base`` only uses GraphBase's public getters, so it works on the unmodified reference object.
"""

import numpy as np

_SCALARS = ("num_layers", "lat_resolution", "lat_offset", "veh_width", "veh_length", "sampled_resolution",
            "vel_decrease_lat", "min_plan_horizon", "plan_horizon_mode", "closed", "virt_goal_node_cost")
_ARRAYS = ("nodes_in_layer", "raceline_index", "s_raceline", "refline", "raceline", "vel_raceline", "normvec",
           "track_width_right", "track_width_left", "alpha", "node_pos", "node_psi", "vgoal_cost", "in_ptr",
           "edge_src", "edge_cost", "edge_len", "edge_coeff", "samp_ptr", "samples", "glob_rl")


NO_VIRT_GOAL_STEP = 1.0e12


def goal_order_cost(raceline_index, nodes_in_layer):
    """Goal costs that make the virtual-goal search pick the node ``GraphBase.search_graph_layer`` picks WITHOUT virtual goal nodes
    (``virt_goal_n=False``, params/ltpl_config_offline.ini:25; GraphBase.py:896-927): there the end layer's nodes are tried one by one
    -- the race-line node first, then the smaller indices down to 0, then the larger ones upwards -- and the first node a path
    reaches wins. A cost of ``NO_VIRT_GOAL_STEP * (position in that order)`` per node, twelve orders of magnitude above any path
    cost, makes "cheapest path + goal cost" choose exactly that node; the path TO it is the same shortest path in both forms. (Where
    a filter has removed a node of the end layer the reference does not skip it: it raises or stops trying -- an error path that is
    not reproduced: removed nodes are simply unreachable here.)"""
    rl = np.asarray(raceline_index, dtype=np.int64)
    K = np.asarray(nodes_in_layer, dtype=np.int64)
    out = []
    for l in range(len(K)):
        n = np.arange(K[l])
        out.append(np.where(n <= rl[l], rl[l] - n, n).astype(np.float64) * NO_VIRT_GOAL_STEP)
    return np.concatenate(out) if out else np.zeros(0)


class Lattice(object):
    """Immutable SoA lattice. All float arrays are C-contiguous float64, all index arrays int32."""

    def __init__(self, **kw):
        for k in _SCALARS + _ARRAYS:
            if k not in kw:
                raise ValueError("Lattice: missing field '%s'" % k)
        self.num_layers = int(kw["num_layers"])
        self.lat_resolution = float(kw["lat_resolution"])
        self.lat_offset = float(kw["lat_offset"])
        self.veh_width = float(kw["veh_width"])
        self.veh_length = float(kw["veh_length"])
        self.sampled_resolution = float(kw["sampled_resolution"])
        self.vel_decrease_lat = float(kw["vel_decrease_lat"])
        self.min_plan_horizon = float(kw["min_plan_horizon"])
        self.plan_horizon_mode = str(kw["plan_horizon_mode"])
        self.closed = bool(kw["closed"])
        self.virt_goal_node_cost = float(kw["virt_goal_node_cost"])

        def f64(a, shape=None):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
            if shape is not None:
                a = a.reshape(shape)
            return a

        def i32(a):
            return np.ascontiguousarray(np.asarray(a, dtype=np.int32))

        L = self.num_layers
        self.nodes_in_layer = i32(kw["nodes_in_layer"])
        self.raceline_index = i32(kw["raceline_index"])
        self.s_raceline = f64(kw["s_raceline"])
        self.refline = f64(kw["refline"], (L, 2))
        self.raceline = f64(kw["raceline"], (L, 2))
        self.vel_raceline = f64(kw["vel_raceline"])
        self.normvec = f64(kw["normvec"], (L, 2))
        self.track_width_right = f64(kw["track_width_right"])
        self.track_width_left = f64(kw["track_width_left"])
        self.alpha = f64(kw["alpha"])
        self.node_pos = f64(kw["node_pos"], (-1, 2))
        self.node_psi = f64(kw["node_psi"])
        self.vgoal_cost = f64(kw["vgoal_cost"])
        self.in_ptr = i32(kw["in_ptr"])
        self.edge_src = i32(kw["edge_src"])
        self.edge_cost = f64(kw["edge_cost"])
        self.edge_len = f64(kw["edge_len"])
        self.edge_coeff = f64(kw["edge_coeff"], (-1, 8))
        self.samp_ptr = i32(kw["samp_ptr"])
        self.samples = f64(kw["samples"], (-1, 5))
        self.glob_rl = f64(kw["glob_rl"], (-1, 5))

        self.layer_off = np.zeros(L + 1, dtype=np.int32)
        self.layer_off[1:] = np.cumsum(self.nodes_in_layer)
        self._check()

    # ---- derived sizes ---------------------------------------------------------------------------------------------
    @property
    def num_nodes(self):
        return int(self.layer_off[-1])

    @property
    def num_edges(self):
        return int(self.edge_src.shape[0])

    @property
    def num_samples(self):
        return int(self.samples.shape[0])

    def _check(self):
        L, V, E, S = self.num_layers, self.num_nodes, self.num_edges, self.num_samples
        assert self.nodes_in_layer.shape == (L,) and self.raceline_index.shape == (L,)
        assert self.s_raceline.shape == (L,) and self.vel_raceline.shape == (L,)
        assert self.node_pos.shape == (V, 2) and self.node_psi.shape == (V,) and self.vgoal_cost.shape == (V,)
        assert self.in_ptr.shape == (V + 1,) and self.in_ptr[0] == 0 and self.in_ptr[-1] == E
        assert self.edge_cost.shape == (E,) and self.edge_len.shape == (E,) and self.edge_coeff.shape == (E, 8)
        assert self.samp_ptr.shape == (E + 1,) and self.samp_ptr[0] == 0 and self.samp_ptr[-1] == S
        assert np.all(np.diff(self.in_ptr) >= 0) and np.all(np.diff(self.samp_ptr) >= 2)
        if E:
            dst = self.edge_dst_gid()
            src_layer = (self.layer_of(dst) - 1) % L
            assert np.all(self.edge_src >= 0) and np.all(self.edge_src < self.nodes_in_layer[src_layer])

    # ---- topology helpers (host side, numpy) -----------------------------------------------------------------------
    def layer_of(self, gid):
        return (np.searchsorted(self.layer_off, np.asarray(gid), side="right") - 1).astype(np.int32)

    def edge_dst_gid(self):
        """Global destination node id of every edge (expansion of the CSC pointer)."""
        return np.repeat(np.arange(self.num_nodes, dtype=np.int32), np.diff(self.in_ptr)).astype(np.int32)

    def edge_endpoints(self):
        """(src_layer, src_node, dst_layer, dst_node) of every edge, int32 arrays of length E."""
        dst = self.edge_dst_gid()
        dl = self.layer_of(dst)
        dn = dst - self.layer_off[dl]
        sl = (dl - 1) % self.num_layers
        return sl.astype(np.int32), self.edge_src.copy(), dl.astype(np.int32), dn.astype(np.int32)

    def find_edge(self, start_layer, start_node, end_layer, end_node):
        """Edge id of (start_layer, start_node) -> (end_layer, end_node) or -1 (cf. GraphBase.get_edge, 444-476)."""
        if end_layer != (start_layer + 1) % self.num_layers:
            return -1
        if not (0 <= end_node < self.nodes_in_layer[end_layer]):
            return -1
        v = int(self.layer_off[end_layer]) + int(end_node)
        lo, hi = int(self.in_ptr[v]), int(self.in_ptr[v + 1])
        k = np.searchsorted(self.edge_src[lo:hi], start_node)
        if k < hi - lo and self.edge_src[lo + k] == start_node:
            return lo + int(k)
        return -1

    def horizon_end_layer(self, start_layer):
        """End layer of the planning range (gen_local_node_template.py:104-133)."""
        import bisect
        if self.plan_horizon_mode == 'distance':
            des_dist = self.s_raceline[start_layer] + self.min_plan_horizon
            if des_dist > self.s_raceline[-1]:
                if self.closed:
                    des_dist -= self.s_raceline[-1]
                else:
                    des_dist = self.s_raceline[-1]
            return bisect.bisect_left(self.s_raceline.tolist(), des_dist)
        if self.plan_horizon_mode == 'layers':
            if self.closed:
                return (start_layer + int(self.min_plan_horizon)) % self.num_layers
            return max((start_layer + int(self.min_plan_horizon)), self.num_layers - 1)
        raise ValueError('Unsupported planning horizon mode "' + self.plan_horizon_mode + '"!')

    def max_horizon(self):
        """(max #layers in a planning range incl. start and end, max #edges, max path samples) over all start layers."""
        L = self.num_layers
        _, _, dl, _ = self.edge_endpoints()
        edges_into = np.bincount(dl, minlength=L)
        n_samp = np.diff(self.samp_ptr)
        max_samp_into = np.zeros(L, dtype=np.int64)
        np.maximum.at(max_samp_into, dl, n_samp)
        best_layers = best_edges = best_pts = 0
        for s in range(L):
            e = self.horizon_end_layer(s)
            dist = e - s if e >= s else L - s + e
            if dist <= 0 or e >= L:
                if not self.closed:
                    continue                # open track: no planning range from the last layer(s)
                dist = L
            layers = [(s + j) % L for j in range(1, dist + 1)]
            best_layers = max(best_layers, dist + 1)
            best_edges = max(best_edges, int(edges_into[layers].sum()))
            best_pts = max(best_pts, int((max_samp_into[layers] - 1).sum()) + 1)
        return best_layers, best_edges, best_pts

    # ---- (de)serialisation -----------------------------------------------------------------------------------------
    def to_dict(self):
        d = {k: getattr(self, k) for k in _ARRAYS}
        for k in _SCALARS:
            d[k] = np.asarray(getattr(self, k))
        return d

    def save(self, path, compressed=True):
        (np.savez_compressed if compressed else np.savez)(path, **self.to_dict())

    @classmethod
    def load(cls, path):
        with np.load(path, allow_pickle=False) as z:
            kw = {k: z[k] for k in _ARRAYS}
            for k in _SCALARS:
                kw[k] = z[k].item()
        return cls(**kw)

    # ---- export from the reference's GraphBase ---------------------------------------------------------------------
    @classmethod
    def from_graph_base(cls, gb):
        """
        Build the SoA lattice from a (reference) ``GraphBase`` through its public API only: ``get_edges`` (648),
        ``get_edge`` (444), ``get_node_info`` (221) and the public attributes of GraphBase.py:93-119.
        """
        virt = bool(getattr(gb, "virt_goal_node", True))
        L = int(gb.num_layers)
        nodes_in_layer = np.array([gb.nodes_in_layer[l] for l in range(L)], dtype=np.int32)
        layer_off = np.zeros(L + 1, dtype=np.int64)
        layer_off[1:] = np.cumsum(nodes_in_layer)
        V = int(layer_off[-1])
        rl_idx = np.array(gb.raceline_index, dtype=np.int32)

        node_pos = np.zeros((V, 2))
        node_psi = np.zeros(V)
        vgoal = np.zeros(V)
        for l in range(L):
            for n in range(nodes_in_layer[l]):
                pos, psi, _, _, _ = gb.get_node_info(layer=l, node_number=n, active_filter=None)
                v = layer_off[l] + n
                node_pos[v] = pos
                node_psi[v] = psi
                # GraphBase.py:188 (same operation order)
                vgoal[v] = abs(int(rl_idx[l]) - n) * gb.lat_resolution * gb.virt_goal_node_cost
        if not virt:
            vgoal = goal_order_cost(rl_idx, nodes_in_layer)

        edges = gb.get_edges()
        recs = []
        for (sl, sn, el, en) in edges:
            if el != (sl + 1) % L:
                raise ValueError("edge (%d,%d)->(%d,%d) does not connect consecutive layers" % (sl, sn, el, en))
            coeff, param, cost, length = gb.get_edge(sl, sn, el, en)
            recs.append((int(layer_off[el] + en), int(sn), np.asarray(coeff, dtype=float).reshape(8),
                         np.asarray(param, dtype=float), float(cost), float(length)))
        recs.sort(key=lambda r: (r[0], r[1]))
        E = len(recs)
        in_ptr = np.zeros(V + 1, dtype=np.int64)
        for r in recs:
            in_ptr[r[0] + 1] += 1
        in_ptr = np.cumsum(in_ptr)
        samp_ptr = np.zeros(E + 1, dtype=np.int64)
        samp_ptr[1:] = np.cumsum([r[3].shape[0] for r in recs])

        return cls(num_layers=L, lat_resolution=gb.lat_resolution, lat_offset=gb.lat_offset, veh_width=gb.veh_width,
                   veh_length=gb.veh_length, sampled_resolution=gb.sampled_resolution,
                   vel_decrease_lat=gb.vel_decrease_lat,
                   min_plan_horizon=getattr(gb, "min_plan_horizon", 200.0),
                   plan_horizon_mode=getattr(gb, "plan_horizon_mode", "distance"),
                   closed=gb.closed, virt_goal_node_cost=gb.virt_goal_node_cost,
                   nodes_in_layer=nodes_in_layer, raceline_index=rl_idx, s_raceline=gb.s_raceline,
                   refline=gb.refline, raceline=gb.raceline, vel_raceline=gb.vel_raceline,
                   normvec=gb.normvec_normalized, track_width_right=gb.track_width_right,
                   track_width_left=gb.track_width_left, alpha=gb.alpha,
                   node_pos=node_pos, node_psi=node_psi, vgoal_cost=vgoal, in_ptr=in_ptr,
                   edge_src=np.array([r[1] for r in recs], dtype=np.int32),
                   edge_cost=np.array([r[4] for r in recs]), edge_len=np.array([r[5] for r in recs]),
                   edge_coeff=np.array([r[2] for r in recs]).reshape(E, 8), samp_ptr=samp_ptr,
                   samples=np.vstack([r[3] for r in recs]), glob_rl=gb.glob_rl)
