"""Doubling method for the sweep: C2 batches whose start layer is FIXED (so every scenario has the same planning range of H layers),
for several H; run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU` and fit
instructions / cycles per scenario against H: the slope is the cost of ONE layer of the sweep (+ its share of path assembly).
    tools/insts_per_layer.py [n_scen] [no_opponents]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphbasedlocaltrajectoryplanner_amd import _capi                        # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice              # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import c2_scenarios    # noqa: E402

lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
no_opp = len(sys.argv) > 2
L = lat.num_layers
H = np.array([(lat.horizon_end_layer(l) - l) % L for l in range(L)])
picks = []
for h in sorted(set(H.tolist())):
    picks.append((h, int(np.nonzero(H == h)[0][0])))
picks = picks[::max(1, len(picks) // 6)]
print("H -> start layer:", picks)
for h, sl in picks:
    scen, _ = c2_scenarios(lat, n, seed=3)
    for sc in scen:
        sc["start_node"] = (sl, int(lat.raceline_index[sl]))
        sc["last_nodes"] = [[(sl + j) % L, int(lat.raceline_index[(sl + j) % L])] for j in range(4)]
        sc["psi_s"] = float(lat.node_psi[lat.layer_off[sl] + lat.raceline_index[sl]])
        if no_opp:
            sc["vehicles"] = []
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    res = hip.new_paths_result(n)
    for _ in range(3):
        hip.plan_paths(batch, res)
    e_h = sum(int(lat.in_ptr[lat.layer_off[(sl + j) % L + 1] if (sl + j) % L + 1 <= L else 0] - lat.in_ptr[lat.layer_off[(sl + j) % L]]) for j in range(1, h + 1))
    print("H=%d sl=%d paths/tick=%.2f pts/path=%.1f edges in range=%d" % (h, sl, res.valid.sum() / n, res.n_pts[res.valid > 0].mean(), e_h))
