"""
ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Binds ``graphbasedlocaltrajectoryplanner_amd.planner.Planner`` to oracle/libltpl_fleet_host.so: the FLEET state machine
(csrc/fleet_core.hpp, the source libltpl_hip.so runs with one wave per planner on device-resident state) compiled with its one-lane
host policy behind the oracle's CPU arithmetic (oracle/fleet_host_shim.cpp). Lets the GPU-less build container replay the
recorded closed loops through exactly the code the device executes.
"""
import ctypes as C
import os
import subprocess

from graphbasedlocaltrajectoryplanner_amd import _capi
from graphbasedlocaltrajectoryplanner_amd.planner import Planner, PlannerConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libltpl_fleet_host.so")


def build(force=False):
    csrc = os.path.join(os.path.dirname(HERE), "graphbasedlocaltrajectoryplanner_amd", "csrc")
    srcs = [os.path.join(HERE, "fleet_host_shim.cpp"), os.path.join(HERE, "ltpl_oracle.c"),
            os.path.join(os.path.dirname(HERE), "include", "ltpl_hip.h")]
    srcs += [os.path.join(csrc, f) for f in ("planner_core.hpp", "planner_host.hpp", "fleet_core.hpp", "fleet_api.hpp")] + [os.path.join(HERE, "oracle_compute.hpp")]
    if not force and os.path.isfile(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(s) for s in srcs):
        return LIB
    # (two processes -- the ranks of a gloo test, pytest-xdist workers -- may find the library stale at the same time: one builds, the
    #  other waits on the lock and finds it fresh)
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if force or not (os.path.isfile(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(s) for s in srcs)):
            subprocess.check_call(["make", "-s", "-C", HERE, "-B", "libltpl_fleet_host.so"])
    return LIB


class HostFleetBackend(object):
    """Stands in for HipBackend when constructing a fleet (``Planner`` with the ``oracle_fleet_`` entry points)."""

    def __init__(self, lattice):
        build()
        self.lib = C.CDLL(LIB)
        self.lattice = lattice
        self.binding = _capi.LatticeBinding(lattice)
        layers, edges, pts = lattice.max_horizon()
        self.max_nodes, self.max_pts = layers, pts
        self.lib.oracle_fleet_create.argtypes = [C.POINTER(_capi.LatticeDesc), C.c_int, C.c_int,
                                                 C.POINTER(PlannerConfig), C.POINTER(C.c_void_p)]

    def planner(self, n_scen=1, **config):
        def create(cfg_ref, handle_ref):
            return self.lib.oracle_fleet_create(C.byref(self.binding.desc), self.max_nodes, self.max_pts, cfg_ref, handle_ref)
        return Planner(self, n_scen=n_scen, lib=self.lib, prefix="oracle_fleet_", create=create, **config)
