"""TEST TOOL: two threads, each with its OWN handle (the ABI's threading contract: a handle is used by one thread at a time, different
handles are independent), creating / ticking / destroying concurrently on the stand-in runtime built with ThreadSanitizer
(FAKEHIP_SAN=thread tools/fakehip/build.sh). Reports data races on the library's process-wide state."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graphbasedlocaltrajectoryplanner_amd import _capi                      # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice            # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import random_scenarios   # noqa: E402

FAKE = os.path.join(ROOT, "tools", "fakehip", "build", "libltpl_hip_fake.so")
errors = []


def worker(k, lat):
    try:
        for rep in range(3):
            hip = _capi.HipBackend(lat, lib_path=FAKE)
            for n in (1, 5, 80):
                scen, vels = random_scenarios(lat, n, seed=10 * k + n)
                batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
                pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
                vel = _capi.TickVelBatch(_capi.VelParamSet(len_veh=lat.veh_length), n, np.full(n, 20.0), np.full(n, 20.0), pos, np.concatenate(vels))
                for _ in range(10):
                    hip.plan_paths(batch)
                    hip.tick_batch(batch, vel)
                hip.vel_profile(_capi.VelParamSet(len_veh=lat.veh_length),
                                [{"mode": _capi.VEL_FB, "kappa": np.zeros(300), "el_lengths": np.ones(299), "loc_gg": np.ones((300, 2)) * 5.0,
                                  "v_start": 10.0, "v_end": 5.0}] * 3)
            hip.close()
    except Exception as e:
        errors.append(repr(e))


lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
ts = [threading.Thread(target=worker, args=(k, lat)) for k in range(2)]
[t.start() for t in ts]
[t.join() for t in ts]
print("threads done, errors:", errors)
sys.exit(1 if errors else 0)
