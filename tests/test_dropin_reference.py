"""
Container-only (needs /root/reference): the drop-in seams against the REAL reference, closed loop.

Run A: the unmodified reference (over oracle/shims, fake clock) drives N ticks of a scenario.
Run B: the same loop after ``install()`` patched seam (1) and seam (2) with this package's host mirrors.
The exported trajectory sets (what ``Graph_LTPL.calc_vel_profile`` returns to the user) must agree tick by tick:
same action keys and ids, [s, x, y, psi, kappa, vx, ax] within 1e-5 relative. There is no GPU in the build container,
so run B binds the mirrors to the oracle's C backend -- this test checks the HOST logic of the mirrors (zone
bookkeeping, constant-segment test, packing / unpacking, VpForwardBackward state) and the install mechanism; the device
arithmetic is checked against the same oracle by the ``-m gpu`` tests.
"""
import os
import warnings
import numpy as np
import pytest

from helpers import assert_close_rel, REL_TOL, ROOT
from oracle import ref_env

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_env.reference_available(), reason="reference tree not present")]

CACHE = os.path.join(ROOT, "oracle", "_cache")


class OracleWithPlanner(object):
    """Backend factory for mode="planner" in the GPU-less container: seam-level calls go to the oracle, the planner entry
    points to the host-logic harness (product state machine + oracle arithmetic)."""

    def __new__(cls, lat):
        from oracle.oracle_lib import OracleBackend
        from oracle.planner_host import HostPlannerBackend
        b = OracleBackend(lat)
        host = HostPlannerBackend(lat)
        b.planner = host.planner
        b._host_planner_backend = host
        return b


def run(scenario, n_ticks, patched, mode="seams"):
    from oracle import ref_scenarios as rs
    from oracle.oracle_lib import OracleBackend
    from graphbasedlocaltrajectoryplanner_amd.install import install, uninstall
    warnings.simplefilter("ignore")
    session = None
    if patched:
        graph_ltpl, clock = ref_env.load_reference()
        session = install(graph_ltpl, backend_factory=OracleBackend if mode == "seams" else OracleWithPlanner, mode=mode,
                          clock=clock)
    try:
        gl, clock, ltpl_obj, gb, path_dict = rs.make_planner(CACHE, clock=session.clock if session is not None else None)
        if scenario == "c2":
            dummies, zones = rs.opponents_c2(gl, 8), rs.ZONE_EXAMPLE
        else:   # a slow opponent in front of a full-width zone: follow / back-off / reduced horizon branches
            from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
            lat = Lattice.from_graph_base(gb)
            zl, zn = [], []
            for layer in (16, 17):
                zl += [layer] * int(lat.nodes_in_layer[layer])
                zn += list(range(int(lat.nodes_in_layer[layer])))
            zones = {'wall_zone': [zl, zn, np.array([[0.0, 0.0], [1.0, 1.0]]), np.array([[0.0, 0.0], [1.0, 1.0]])]}
            dummies = [gl.testing_tools.src.objectlist_dummy.ObjectlistDummy(dynamic=True, vel_scale=0.15, s0=180.0)]
        exported = rs.run_loop(gl, clock, ltpl_obj, path_dict, n_ticks=n_ticks, dt=0.05, dummies=dummies, zones=zones)
        if patched:
            assert session.current is not None, "install(): the patched seams were never reached"
        return exported
    finally:
        if session is not None:
            uninstall(session)


@pytest.mark.parametrize("scenario,n_ticks,mode", [("c2", 160, "seams"), ("zonewall", 140, "seams"),
                                                   ("c2", 400, "planner"), ("zonewall", 200, "planner")])
def test_closed_loop_matches_unmodified_reference(scenario, n_ticks, mode):
    ref = run(scenario, n_ticks, patched=False)
    got = run(scenario, n_ticks, patched=True, mode=mode)
    assert len(ref) == len(got) == n_ticks
    seen = set()
    for t, (a, b) in enumerate(zip(ref, got)):
        assert a['sel_action'] == b['sel_action'], "tick %d" % t
        assert list(a['traj'].keys()) == list(b['traj'].keys()), "tick %d: %s vs %s" % (t, a['traj'].keys(), b['traj'].keys())
        assert a['traj_id'] == b['traj_id']
        assert_close_rel(b['pos_est'], a['pos_est'], what="tick %d pos_est" % t)
        for k in a['traj']:
            ta, tb = a['traj'][k], b['traj'][k]
            assert ta.shape == tb.shape, "tick %d %s" % (t, k)
            for col, name in enumerate(("s", "x", "y", "psi", "kappa", "vx", "ax")):
                if name == "psi":
                    d = np.abs(np.mod(tb[:, col] - ta[:, col] + np.pi, 2 * np.pi) - np.pi)
                    assert float(d.max()) <= REL_TOL * np.pi, "tick %d %s psi" % (t, k)
                elif name == "ax":
                    scale = max(float(np.max(np.abs(ta[:, 5]))) ** 2 / 2.0, 5.0)
                    assert float(np.max(np.abs(tb[:, col] - ta[:, col]))) <= 1e-5 * scale, "tick %d %s ax" % (t, k)
                else:
                    assert_close_rel(tb[:, col], ta[:, col], what="tick %d %s %s" % (t, k, name))
            seen.add(k)
    if scenario == "c2":
        assert "follow" in seen and seen & {"left", "right"}


@pytest.mark.parametrize("launcher_mode", ["planner", "seams"])
def test_launcher_runs_unmodified_std_example(tmp_path, monkeypatch, launcher_mode):
    """`python -m graphbasedlocaltrajectoryplanner_amd.run --ticks N main_std_example.py` on a scratch checkout (the
    example writes logs / the graph cache next to itself, and /root/reference is read-only). The script itself is the
    reference's file, byte for byte; only install()'s default backend factory is swapped for the oracle (no GPU here)."""
    import shutil
    import sys
    from oracle.oracle_lib import OracleBackend
    import graphbasedlocaltrajectoryplanner_amd.install as inst
    import graphbasedlocaltrajectoryplanner_amd.run as launcher
    ref = ref_env.REFERENCE_ROOT
    shutil.copy(os.path.join(ref, "main_std_example.py"), tmp_path / "main_std_example.py")
    shutil.copytree(os.path.join(ref, "params"), tmp_path / "params")
    shutil.copytree(os.path.join(ref, "inputs"), tmp_path / "inputs")
    os.symlink(os.path.join(ref, "graph_ltpl"), tmp_path / "graph_ltpl")
    cached = os.path.join(CACHE, "stored_graph_monteblanco.pckl")
    if os.path.isfile(cached):
        shutil.copy(cached, tmp_path / "inputs" / "stored_graph.pckl")
    monkeypatch.setenv("MPLBACKEND", "Agg")
    real_install = inst.install
    sessions = []

    def install_with_oracle(graph_ltpl, backend_factory=None, device=-1, mode="seams", clock=None):
        sessions.append(real_install(graph_ltpl, backend_factory=OracleBackend if mode == "seams" else OracleWithPlanner,
                                     device=device, mode=mode, clock=clock))
        return sessions[-1]
    monkeypatch.setattr(inst, "install", install_with_oracle)
    saved_path, saved_argv = list(sys.path), list(sys.argv)
    try:
        with pytest.raises(SystemExit) as exc:
            launcher.main(["--ticks", "40", "--mode", launcher_mode, "--extra-path", os.path.join(ROOT, "oracle", "shims"),
                           str(tmp_path / "main_std_example.py")])
        assert exc.value.code == 0
        assert sessions and sessions[0].current is not None
        if launcher_mode == "seams":
            gen = sessions[0].current[3]
            assert gen.last_result is not None and int(gen.last_result.n_actions[0]) >= 1
    finally:
        sys.path[:], sys.argv[:] = saved_path, saved_argv
        if sessions:
            inst.uninstall(sessions[0])
        import matplotlib.pyplot as plt
        plt.close("all")
