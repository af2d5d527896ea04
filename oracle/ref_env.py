"""
ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the UNMODIFIED reference package ``graph_ltpl`` from /root/reference (only present in the build container,
never on the GPU box) on top of the three dependency shims in oracle/shims/ (igraph, trajectory_planning_helpers,
zmq), applies the two NumPy aliases the reference still uses (``np.Inf`` main_online_path_gen.py:96, ``np.object``
main_offline_callback.py:160) and installs a deterministic fake clock in every reference module that reads wall time
(OnlineTrajectoryHandler.py:353-354,672; ObjectListInterface.py:87,143; objectlist_dummy.py:148-149).

Used by oracle/gen_golden.py (fixture generation) and by the container-only tests that compare the host mirror
against the real reference. Nothing here is reachable from bench.py's timed region or from the C-ABI library.
"""

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("LTPL_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "graph_ltpl", "Graph_LTPL.py"))


class FakeClock(object):
    """Counter clock: ``time()`` returns the current value; ``advance(dt)`` moves it. Stands in for ``time`` module."""

    def __init__(self, t0: float = 1.0e6):
        self.now = float(t0)

    def time(self) -> float:
        return self.now

    def advance(self, dt: float) -> None:
        self.now += float(dt)

    # the reference only calls time.time(); sleep is provided for completeness
    def sleep(self, dt: float) -> None:
        self.advance(dt)


def load_reference(clock: FakeClock = None):
    """Import the reference over the shims. Returns (graph_ltpl module, clock)."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s (it only exists in the build container)" % REFERENCE_ROOT)

    import numpy as np
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    if not hasattr(np, "object"):
        np.object = object

    if _SHIMS not in sys.path:
        sys.path.insert(0, _SHIMS)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)

    import graph_ltpl  # noqa: E402  (the unmodified reference)

    if clock is None:
        clock = FakeClock()
    fake_time = types.SimpleNamespace(time=clock.time, sleep=clock.sleep)
    for mod in (graph_ltpl.online_graph.src.OnlineTrajectoryHandler,
                graph_ltpl.data_objects.ObjectListInterface,
                graph_ltpl.testing_tools.src.objectlist_dummy,
                graph_ltpl.helper_funcs.src.calc_vel_profile_follow):
        mod.time = fake_time
    return graph_ltpl, clock


def default_path_dict(cache_dir: str, track: str = "monteblanco") -> dict:
    """path_dict as in main_std_example.py:33-41, with the graph cache redirected outside the read-only reference."""
    os.makedirs(cache_dir, exist_ok=True)
    return {'globtraj_input_path': REFERENCE_ROOT + "/inputs/traj_ltpl_cl/traj_ltpl_cl_" + track + ".csv",
            'graph_store_path': os.path.join(cache_dir, "stored_graph_" + track + ".pckl"),
            'ltpl_offline_param_path': REFERENCE_ROOT + "/params/ltpl_config_offline.ini",
            'ltpl_online_param_path': REFERENCE_ROOT + "/params/ltpl_config_online.ini"}
