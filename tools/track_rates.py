"""Per-track throughput of the tick pipeline on the six race line files the reference ships (inputs/traj_ltpl_cl): for every track the
C2-style workload (8 race-line followers with a 0.2 s prediction, previous solution, sample zone where the lattice has those layers) on
ITS lattice -- ticks/s of the resident pipeline, path-kernel time, the plan class the library chose for the one-wave batch kernel with
its register count and waves per SIMD, and the rate per edge of the planning range relative to Monteblanco.

    python tools/track_rates.py [batch] > profiles/<tag>_tracks.txt          (GPU box; berlin / modena lattices are rebuilt by the offline build)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                                    # noqa: E402
import bench                                                                          # noqa: E402
import __graft_entry__ as ge                                                          # noqa: E402
from graphbasedlocaltrajectoryplanner_amd import _capi                               # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice                     # noqa: E402


def lattice_of(track):
    if track == "monteblanco":
        return Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    import test_other_tracks as T
    return T.lattice_of(track)


def plan_class(lat, hmax):
    kmax, L = int(np.max(lat.nodes_in_layer)), lat.num_layers
    for name, kp, hm in (("PlanFx<32,32,1>", 32, 32), ("PlanFx<32,40,1>", 32, 40), ("PlanFx<48,32,1>", 48, 32)):
        if kmax <= kp and hmax + 1 <= hm and L >= hm:
            return name
    return "PlanRt"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    res_all = ge.kernel_resources(ge.HIP_LIB)
    rows, base = [], None
    for track in ("monteblanco", "zalazone", "millbrook", "lvms", "berlin", "modena"):
        lat = lattice_of(track)
        hip = _capi.HipBackend(lat)
        scen, batch, vel = bench.make_batch(lat, n, seed=1)
        hip.batch_upload(batch, vel)
        hip.batch_run(reps=5, timed=False)
        reps = 30
        t0 = time.perf_counter(); hip.batch_run(reps=reps, timed=True); el = time.perf_counter() - t0
        paths_ms = hip.batch_last_paths_ms()
        res, _ = hip.batch_download()
        sl = np.asarray(batch.start_layer[:n]); elr = np.asarray(res.end_layer[:n]); L = lat.num_layers
        H = np.where(elr >= sl, elr - sl, L - sl + elr)
        deg = np.diff(np.asarray(lat.in_ptr)); lo = np.asarray(lat.layer_off)
        ne = np.array([deg[lo[l]:lo[l + 1]].sum() for l in range(L)])
        cum = np.concatenate(([0], np.cumsum(np.concatenate((ne, ne)))))
        e_h = float(np.mean(cum[sl + 1 + H] - cum[sl + 1]))                     # edges of the planning range, mean over the batch
        pc = plan_class(lat, hip.caps.max_path_nodes)
        key = {"PlanFx<32,32,1>": "PlanFxILi32ELi32ELi1", "PlanFx<32,40,1>": "PlanFxILi32ELi40ELi1", "PlanFx<48,32,1>": "PlanFxILi48ELi32ELi1",
               "PlanRt": "k_pathsILi1E6PlanRtE"}[pc]
        kr = [v for k, v in res_all.items() if "k_pathsILi1E" in k and key in k]
        vg = kr[0]["vgpr_count"] if kr else -1
        rate = n * reps / el
        row = dict(track=track, layers=L, kmax=int(np.max(lat.nodes_in_layer)), hmax=int(hip.caps.max_path_nodes), edges=int(lat.num_edges),
                   ticks_per_s=rate, k_paths_ms=paths_ms, plan=pc, vgprs=vg, waves_per_simd=(512 // max(vg, 1)) if vg > 0 else -1,
                   paths_per_tick=float(res.valid.sum()) / n, horizon_edges=e_h, edge_rate=rate * e_h)
        if base is None:
            base = row["edge_rate"]
        row["edge_rate_vs_monteblanco"] = row["edge_rate"] / base
        rows.append(row)
        hip.close()
        print("%-12s L %4d kmax %3d hmax %3d | %6.2f M ticks/s  k_paths %.3f ms / %d | %-16s %3d VGPRs %d waves/SIMD | %.2f paths/tick, %5.0f horizon "
              "edges, per-edge rate vs monteblanco %.2f" % (track, L, row["kmax"], row["hmax"], rate / 1e6, paths_ms, n, pc, vg, row["waves_per_simd"],
                                                            row["paths_per_tick"], e_h, row["edge_rate_vs_monteblanco"]), flush=True)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
