"""Seeded synthetic scenario generators shared by tests and bench.py (no reference needed)."""
import numpy as np


def raceline_state(lat, s):
    """Position / heading (0 = north) / speed on the fine global race line at arc length s (wraps)."""
    g = lat.glob_rl
    s = np.mod(s, g[-1, 0])
    x = np.interp(s, g[:, 0], g[:, 1])
    y = np.interp(s, g[:, 0], g[:, 2])
    i = np.minimum(np.searchsorted(g[:, 0], s, side="right") - 1, g.shape[0] - 2)
    psi = np.arctan2(g[i + 1, 2] - g[i, 2], g[i + 1, 1] - g[i, 1]) - np.pi / 2
    v = np.interp(s, g[:, 0], g[:, 4])
    return x, y, psi, v


def random_scenarios(lat, n, seed, n_veh=8, lateral=True, zone_prob=0.3, last_prob=0.7):
    """
    Independent planning problems on a lattice in the spirit of SURVEY.md §8d C2/C4: ego start node on a random layer,
    ``n_veh`` vehicles placed around the track (several of them within the planning horizon), 0.2 s constant-velocity
    prediction each (ObjectListInterface.py:117-127), optional zone, optional previous-solution cost discount and
    random constant-segment flags. Returns (list of scenario dicts, list of vehicle speed arrays).
    """
    rng = np.random.default_rng(seed)
    L = lat.num_layers
    track_len = float(lat.glob_rl[-1, 0])
    scen, vels = [], []
    for _ in range(n):
        sl = int(rng.integers(0, L))
        sn = int(np.clip(lat.raceline_index[sl] + rng.integers(-3, 4), 0, lat.nodes_in_layer[sl] - 1))
        s_ego = float(lat.s_raceline[sl])
        vehicles, vv = [], []
        for k in range(n_veh):
            if k < max(1, n_veh // 2):
                s_obj = s_ego + rng.uniform(15.0, 280.0)
            else:
                s_obj = rng.uniform(0.0, track_len)
            x, y, psi, v = raceline_state(lat, s_obj)
            if lateral:
                i = int(np.argmin((lat.refline[:, 0] - x) ** 2 + (lat.refline[:, 1] - y) ** 2))
                off = rng.uniform(-2.0, 2.0)
                x, y = x + lat.normvec[i, 0] * off, y + lat.normvec[i, 1] * off
            v = float(v) * rng.uniform(0.2, 0.6)
            dt = 0.2
            pred = np.array([[x - np.sin(psi) * v * dt, y + np.cos(psi) * v * dt]])
            vehicles.append((2.5, np.vstack((np.array([[x, y]]), pred))))
            vv.append(v)
        zone = []
        if rng.random() < zone_prob:
            zl = (sl + int(rng.integers(3, 12))) % L
            k0 = int(rng.integers(0, max(1, lat.nodes_in_layer[zl] - 4)))
            for dl in range(2):
                l2 = (zl + dl) % L
                for nn in range(k0, min(k0 + 5, lat.nodes_in_layer[l2])):
                    zone.append(int(lat.layer_off[l2]) + nn)
        last_nodes = None
        if rng.random() < last_prob:
            # a plausible previous solution: follow existing edges for 4 layers
            last_nodes = [[sl, sn]]
            cur = sn
            for j in range(1, 5):
                l2 = (sl + j) % L
                cands = [nn for nn in range(lat.nodes_in_layer[l2])
                         if lat.find_edge((l2 - 1) % L, cur, l2, nn) >= 0]
                if not cands:
                    break
                cur = int(cands[int(rng.integers(0, len(cands)))])
                last_nodes.append([l2, cur])
        r = rng.random()
        sc = {"start_node": (sl, sn), "action_sets": True, "vehicles": vehicles, "zone_gids": zone,
              "last_nodes": last_nodes, "obj_in_const": False, "obj_besides": False, "last_action": None,
              "const_closest": None, "psi_s": None}
        if r < 0.25:
            sc["obj_besides"] = True
            sc["const_closest"] = 0
            sc["last_action"] = [None, "left", "right", "straight"][int(rng.integers(0, 4))]
            sc["obj_in_const"] = bool(rng.random() < 0.3)
        if rng.random() < 0.5:
            sc["psi_s"] = float(lat.node_psi[lat.layer_off[sl] + sn] + rng.uniform(-0.05, 0.05))
        scen.append(sc)
        vels.append(np.array(vv))
    return scen, vels


SAMPLE_ZONE = ([64] * 7 + [65] * 7 + [66] * 7, list(range(7)) * 3)     # main_std_example.py:90-91


def c2_scenarios(lat, n, seed=1, n_opp=8, lead_gap=None):
    """
    Batch of independent C2 scenarios: the std-example opponent set -- ``n_opp`` race-line
    followers at s0_k = 250 + 280 k with vel_scale_k = 0.30 + 0.05 (k mod 4), 0.2 s constant-velocity prediction
    (O = 2 n_opp obstacle positions) -- phase-shifted by rng.uniform(0, track length); ego start layer
    rng.integers(0, L) on the race-line node; the sample zone of main_std_example.py; previous solution = race-line
    nodes (cost discount w_last_edges), constant-segment heading = node heading. Returns (scenarios, vehicle speeds).
    ``lead_gap=(lo, hi)``: the phase shift is chosen such that opponent 0 drives rng.uniform(lo, hi) metres ahead of the ego
    (every scenario then has an object inside the planning horizon: the [follow, left, right] template is live).
    """
    rng = np.random.default_rng(seed)
    L = lat.num_layers
    track_len = float(lat.glob_rl[-1, 0])
    zone = [int(lat.layer_off[l]) + nn for l, nn in zip(*SAMPLE_ZONE)
            if l < L and nn < lat.nodes_in_layer[l]]
    scen, vels = [], []
    for _ in range(n):
        shift = rng.uniform(0.0, track_len)
        sl = int(rng.integers(0, L))
        sn = int(lat.raceline_index[sl])
        if lead_gap is not None:
            shift = float(lat.s_raceline[sl]) + rng.uniform(lead_gap[0], lead_gap[1]) - 250.0
        vehicles, vv = [], []
        for k in range(n_opp):
            s_k = 250.0 + 280.0 * k + shift
            x, y, psi, v = raceline_state(lat, s_k)
            v = float(v) * (0.30 + 0.05 * (k % 4))
            dt = 0.2
            pred = np.array([[x - np.sin(psi) * v * dt, y + np.cos(psi) * v * dt]])
            vehicles.append((2.5, np.vstack((np.array([[x, y]]), pred))))
            vv.append(v)
        last_nodes = [[(sl + j) % L, int(lat.raceline_index[(sl + j) % L])] for j in range(4)]
        scen.append({"start_node": (sl, sn), "action_sets": True, "vehicles": vehicles, "zone_gids": zone,
                     "last_nodes": last_nodes, "obj_in_const": False, "obj_besides": False, "last_action": None,
                     "const_closest": None, "psi_s": float(lat.node_psi[lat.layer_off[sl] + sn])})
        vels.append(np.array(vv))
    return scen, vels
