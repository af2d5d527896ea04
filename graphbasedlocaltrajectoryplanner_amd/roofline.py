"""
Algorithmic-byte model of one planning tick evaluated on ACTUAL counts of a batch.

    B_tick = B_mask + B_sweep + B_path + B_vel          (fp64 = 8 B, int32 = 4 B)
    B_mask  = sum over obstacle positions of  16 N_L                         reference-line scan (x, y)
              + for gated positions: sum over their window transitions inside the planning range of
                16 n_samples + 1 n_edges                                      sample (x, y) reads, 1 flag per edge
    B_sweep = sum over executed sweeps of  13 E_h + 12 V_h + 8 K_end          cost 8 + src 4 + mask 1 per edge,
                                                                              dist 8 + parent 4 per node, goal costs
    B_path  = sum over produced paths of  80 n_p + 64 L_p                     5 fp64 in + 5 out per sample, coeffs
    B_vel   = sum over produced paths of  48 n_p                              kappa, len, ax, ay in; vx, ax out

These are the bytes an ideal implementation has to touch, not the traffic the kernel generates (the lattice is L2 /
Infinity-Cache resident); bench.py divides them by the measured kernel time to obtain ``roofline.achieved``.
"""
import numpy as np

from . import _capi


def algorithmic_bytes(lat, batch: _capi.PathsBatch, res: _capi.PathsResult):
    L = lat.num_layers
    _, _, dl, _ = lat.edge_endpoints()
    edges_into = np.bincount(dl, minlength=L).astype(np.int64)
    samp_into = np.zeros(L, dtype=np.int64)
    np.add.at(samp_into, dl, np.diff(lat.samp_ptr).astype(np.int64))
    K = lat.nodes_in_layer.astype(np.int64)
    ref = lat.refline
    b_mask = b_sweep = b_path = b_vel = 0
    for s in range(batch.n_scen):
        sl, el = int(batch.start_layer[s]), int(res.end_layer[s])
        H = el - sl if el >= sl else L - sl + el
        layers = [(sl + j) % L for j in range(H + 1)]
        E_h = int(edges_into[layers[1:]].sum())
        V_h = int(K[layers].sum())
        # obstacle positions
        v0, v1 = int(batch.veh_off[s]), int(batch.veh_off[s + 1])
        p0, p1 = int(batch.pos_off[v0]), int(batch.pos_off[v1])
        for p in range(p0, p1):
            b_mask += 16 * L
            ol = int(np.argmin((ref[:, 0] - batch.pos_x[p]) ** 2 + (ref[:, 1] - batch.pos_y[p]) ** 2))
            gate = (sl - 1 <= ol <= el + 1) or (sl > el and (sl - 1 <= ol or ol <= el + 1))
            if not gate:
                continue
            for second in (0, 1):
                if second and ol > L - 2:
                    continue
                b = (ol + second) % L
                jb = (b - sl) % L
                if 1 <= jb <= H:
                    b_mask += 16 * int(samp_into[b]) + int(edges_into[b])
        # sweeps actually executed: [straight] -> 1, constant-segment template -> 2, [follow, left, right] -> 3
        n_act = int(res.n_actions[s])
        besides = bool(batch.flags[s] & (_capi.FLAG_OBJ_IN_CONST | _capi.FLAG_OBJ_BESIDES))
        n_sweeps = 1 if n_act == 1 and not besides else (2 if besides else 3)
        if besides and n_act == 1:
            n_sweeps = 1
        b_sweep += n_sweeps * (13 * E_h + 12 * V_h + 8 * int(K[el]))
        for a in range(n_act):
            if res.valid[s, a]:
                n_p, L_p = int(res.n_pts[s, a]), int(res.n_nodes[s, a]) - 1
                b_path += 80 * n_p + 64 * L_p
                b_vel += 48 * n_p
    return {"mask": b_mask, "sweep": b_sweep, "path": b_path, "vel": b_vel,
            "total": b_mask + b_sweep + b_path + b_vel}
