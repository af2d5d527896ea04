"""Debug helper (GPU box): C3 batch, HIP vs oracle, print the scenarios whose integer outputs differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice, scattered_obstacle_scenarios
from oracle.oracle_lib import OracleBackend

FIELDS = ("end_layer", "closest_obj_index", "closest_obj_node", "n_actions", "action_id", "valid", "reduced",
          "goal_layer", "n_nodes", "n_pts", "n_ties")
lat = c3_lattice()
hip = _capi.HipBackend(lat)
orc = OracleBackend(lat)
scen, vels = scattered_obstacle_scenarios(lat, 256, n_obj=32, seed=0)
batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
ref = orc.plan_paths(batch)
if os.environ.get("LTPL_VGPR_POISON"):
    import ctypes
    vp = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libvgpr_poison.so"))
    print("vgpr poison", vp.vgpr_poison(ctypes.c_uint(int(os.environ["LTPL_VGPR_POISON"], 0))))
res = hip.plan_paths(batch)
bad = {}
for name in FIELDS:
    a, b = getattr(res, name), getattr(ref, name)
    rows = [s for s in range(256) if not np.array_equal(a[s], b[s])]
    if rows:
        bad[name] = rows
        print("field", name, "differs in scenarios", rows[:20], "(%d)" % len(rows))
nodes_bad = []
for s in range(256):
    for a in range(3):
        if res.valid[s, a] and ref.valid[s, a]:
            nn = int(ref.n_nodes[s, a])
            if not np.array_equal(res.nodes[s, a, :nn], ref.nodes[s, a, :nn]):
                nodes_bad.append((s, a))
print("node sequences differ:", nodes_bad[:20], "(%d)" % len(nodes_bad))
first = sorted(set(sum(bad.values(), [])) | set(s for s, _ in nodes_bad))[:3]
for s in first:
    print("scen", s, "start", scen[s]["start_node"], "closest", ref.closest_obj_node[s], "valid", res.valid[s], ref.valid[s],
          "n_pts", res.n_pts[s], ref.n_pts[s], "ties", res.n_ties[s], ref.n_ties[s])
    for a in range(3):
        if ref.valid[s, a]:
            nn = int(ref.n_nodes[s, a])
            print("   slot", a, "hip", res.nodes[s, a, :nn].tolist())
            print("   slot", a, "ref", ref.nodes[s, a, :nn].tolist())
            print("   slot", a, "hip idx", res.node_idx[s, a, :nn].tolist())
            print("   slot", a, "ref idx", ref.node_idx[s, a, :nn].tolist())
    r1 = hip.plan_paths(_capi.PathsBatch([scen[s]], w_last_edges=[0.0, 0.5, 0.8]))
    print("   single (NW=4): n_pts", r1.n_pts[0], "valid", r1.valid[0])
if not bad and not nodes_bad:
    print("ALL EQUAL")
