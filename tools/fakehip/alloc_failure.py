"""TEST TOOL: "no C++ exception crosses the ABI". The library's host code built against the stand-in runtime WITHOUT sanitizers
(FAKEHIP_SAN=none tools/fakehip/build.sh), the process's address space capped just above its current size (RLIMIT_AS), then entry
points that allocate host containers are called: the failed allocation must come back as a status code with a message
(LTPL_ERR_EXCEPTION = 6, or the runtime's own allocation error), never as an abort, and the handle must keep working afterwards."""
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graphbasedlocaltrajectoryplanner_amd import _capi                      # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice            # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.planner import Planner            # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import random_scenarios   # noqa: E402

FAKE = os.path.join(ROOT, "tools", "fakehip", "build_plain", "libltpl_hip_fake.so")


def vm_bytes():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[0]) * 4096


def capped(slack, fn):
    resource.setrlimit(resource.RLIMIT_AS, (vm_bytes() + slack, resource.RLIM_INFINITY))
    try:
        fn()
        return None
    except _capi.BackendError as e:
        return str(e)
    except MemoryError:
        return "python"                                         # the interpreter ran out first: says nothing about the library
    finally:
        resource.setrlimit(resource.RLIMIT_AS, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
hip = _capi.HipBackend(lat, lib_path=FAKE)
scen, vels = random_scenarios(lat, 3000, seed=1)
batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
res = hip.new_paths_result(3000)
seen = set()
# (the planner batch is large on purpose: memory the allocator still holds from ltpl_create would otherwise serve a small one without
#  touching the capped address space)
for slack in (0, 1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
    for what, fn in (("planner_create", lambda: Planner(hip, 40000).close()), ("plan_paths", lambda: hip.plan_paths(batch, res))):
        msg = capped(slack, fn)
        if msg and msg != "python":
            kind = "exception" if "C++ exception caught at the ABI" in msg else "runtime"
            seen.add((what, kind))
            print("slack %4d KiB  %-15s -> %s" % (slack >> 10, what, msg[:120]))
ok = hip.plan_paths(batch, res)                                 # the handle still works
print("after the limit is lifted: plan_paths returns, n_actions[:3] =", ok.n_actions[:3].tolist())
assert any(kind == "exception" for _, kind in seen), "no C++ allocation failure was provoked: %s" % seen
print("alloc-failure check OK:", sorted(seen))
