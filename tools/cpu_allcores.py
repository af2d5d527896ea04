"""CPU baseline on all host cores (SURVEY section 8d asks for 1 core AND all cores): the plain-C oracle (oracle_tick_batch) in
one process per core, every process on its own shard of the same C2 scenarios, ~10 s each. Prints one JSON line."""
import json
import multiprocessing as mp
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, n_proc, n_per, q):
    import bench
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    from oracle.oracle_lib import OracleBackend
    lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    scen, batch, vel = bench.make_batch(lat, n_per, seed=1 + rank)
    orc = OracleBackend(lat)
    orc.tick_batch(batch, vel)
    q.put(("ready", rank))
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 10.0:
        orc.tick_batch(batch, vel); reps += 1
    q.put(("done", n_per * reps / (time.perf_counter() - t0)))


if __name__ == "__main__":
    n_proc = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    q = mp.Queue()
    procs = [mp.Process(target=worker, args=(r, n_proc, 512, q)) for r in range(n_proc)]
    for p in procs:
        p.start()
    rates = []
    while len(rates) < n_proc:
        kind, val = q.get()
        if kind == "done":
            rates.append(val)
    for p in procs:
        p.join()
    print(json.dumps({"cpu_all_cores": {"value": float(np.sum(rates)), "unit": "ticks/s", "cores": n_proc, "kind": "port",
                                        "per_process_min": float(np.min(rates)), "per_process_max": float(np.max(rates)),
                                        "sample": "%d processes x 512 C2 scenarios, oracle_tick_batch, ~10 s each" % n_proc}}))
