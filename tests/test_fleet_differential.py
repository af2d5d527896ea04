"""
CPU: differential closed loops of the fleet's state machine (csrc/fleet_core.hpp, one-lane host build) against the product's host planner
(csrc/planner_core.hpp), both over the oracle's arithmetic: same inputs -> IDENTICAL outputs, call by call, on seeded traffic that reaches
branches the recordings visit rarely (tests/fleet_differential.py). The host planner is the one pinned to the reference's recordings.
"""
import pytest

from fleet_differential import drive


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_fleet_and_host_planner_agree_call_by_call(monteblanco, seed):
    from oracle.fleet_host import HostFleetBackend
    from oracle.planner_host import HostPlannerBackend
    A, B = HostPlannerBackend(monteblanco).planner(1), HostFleetBackend(monteblanco).planner(2)
    st = drive(monteblanco, A, B, seed, 500, exact=True, scen_b=1)
    assert st['ticks'] >= 300, st


def test_the_driver_reaches_the_rare_branches(monteblanco):
    from oracle.planner_host import HostPlannerBackend
    total = {'keys': set(), 'emergency_prev': 0, 'red_len': 0, 'errors': 0, 'dropped': 0, 'restarts': 0}
    for seed in (1, 2, 3, 4, 5, 6):
        A, B = HostPlannerBackend(monteblanco).planner(1), HostPlannerBackend(monteblanco).planner(1)
        st = drive(monteblanco, A, B, seed, 500)
        total['keys'] |= st['keys']
        for k in ('emergency_prev', 'red_len', 'errors', 'dropped', 'restarts'):
            total[k] += st[k]
    assert {'straight', 'follow', 'left', 'right', 'emergency'} <= total['keys'], total
    assert total['emergency_prev'] > 10 and total['red_len'] > 100 and total['dropped'] > 10, total


@pytest.mark.gpu
@pytest.mark.parametrize("seed,follow_form", [(1, None), (3, None), (4, None), (1, "0"), (4, "0")])
def test_fleet_and_host_planner_agree_on_the_device(monteblanco, monkeypatch, seed, follow_form):
    """The same on the MI355X: ltpl_fleet_* (device-resident state, 70 planners = one-wave batch path kernel) against ltpl_planner_* on the
    same handle. Not bit-equal (the fleet solves forward-backward jobs one lane per job as w = v^2 recurrences, the host planner goes through
    the wave-per-job kernel): node lists, indices, keys and ids identical, arrays to 1e-5 relative, vx sample by sample. follow_form "0":
    LTPL_FLEET_FOLLOW_WAVES=0, the lane form of the FOLLOW jobs that fleets of >= 12 288 planners run by default."""
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    if follow_form is not None:
        monkeypatch.setenv("LTPL_FLEET_FOLLOW_WAVES", follow_form)
    hip = _capi.HipBackend(monteblanco)
    A, B = Planner(hip, 1), Fleet(hip, 70)
    st = drive(monteblanco, A, B, seed, 300, exact=False, scen_b=69)
    assert st['ticks'] >= 200, st
    A.close(); B.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_friction_rows_then_constants_with_a_loss_of_grip(monteblanco, seed):
    """Ticks with friction rows, then ticks without, the car losing grip at the switch: the backup brake plan is solved on the rows the
    PREVIOUS tick stored (OTH.py:963-968) in a call that carries none. Host state machines against each other (exact)."""
    from oracle.fleet_host import HostFleetBackend
    from oracle.planner_host import HostPlannerBackend
    A, B = HostPlannerBackend(monteblanco).planner(1), HostFleetBackend(monteblanco).planner(2)
    st = drive(monteblanco, A, B, seed, 260, exact=True, scen_b=1, gg_phases=True)
    assert st['ticks'] >= 150 and st.get('gg_row_ticks', 0) >= 60, st


@pytest.mark.gpu
@pytest.mark.parametrize("seed,follow_form", [(1, None), (2, None), (2, "0")])
def test_friction_rows_then_constants_on_the_device(monteblanco, monkeypatch, seed, follow_form):
    """The same on the MI355X: round 4 chose the friction-row form of the brake-job kernel by whether the CURRENT call carried rows, so a
    backup plan with stored rows was solved with its first row's limits once the rows stopped coming (advisor finding, round 4); the fleet
    now launches the rows form for backup / emergency jobs from the first call with rows on (fleet_dev.hpp, `seen_gg`)."""
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    if follow_form is not None:                      # (the lane form of the follow jobs in the phases without rows, wave-per-job in those with)
        monkeypatch.setenv("LTPL_FLEET_FOLLOW_WAVES", follow_form)
    hip = _capi.HipBackend(monteblanco)
    A, B = Planner(hip, 1), Fleet(hip, 70)
    st = drive(monteblanco, A, B, seed, 260, exact=False, scen_b=69, gg_phases=True)
    assert st['ticks'] >= 150 and st.get('gg_row_ticks', 0) >= 60, st
    A.close(); B.close()


@pytest.mark.parametrize("track", ["zalazone", "millbrook", "lvms"])
def test_fleet_and_host_planner_agree_on_other_tracks(track):
    """The same differential loop on lattices of other plan classes (runtime LDS plan, one-node layers, a long oval)."""
    import os
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    from oracle.fleet_host import HostFleetBackend
    from oracle.planner_host import HostPlannerBackend
    lat = Lattice.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", track + "_lattice.npz"))
    for seed in (2, 3):
        st = drive(lat, HostPlannerBackend(lat).planner(1), HostFleetBackend(lat).planner(1), seed, 300, exact=True)
        assert st['ticks'] >= 200, st


@pytest.mark.parametrize("horizon", [100.0, 300.0])
def test_high_resolution_lattice_with_a_long_horizon(horizon):
    """C5: 0.5 m layer spacing, 202 / 602 layers of planning range -- more path NODES (2 x 602 + 8) than a block has trajectory rows per
    scratch array; the scratch rows of a planner block are sized for both since round 4 (fleet::Dims::SR; before, such lattices were
    refused at create time)."""
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c5_lattice
    from oracle.fleet_host import HostFleetBackend
    from oracle.planner_host import HostPlannerBackend
    lat = c5_lattice(horizon=horizon)
    keys = set()
    for seed in (2, 3, 5):
        st = drive(lat, HostPlannerBackend(lat).planner(1), HostFleetBackend(lat).planner(2), seed, 60, exact=True, scen_b=1)
        assert st['ticks'] >= 40, st
        keys |= st['keys']
    assert "straight" in keys, keys


@pytest.mark.gpu
def test_long_horizon_fleet_on_the_device():
    """C5 as BASELINE specifies it (600 layers of planning range: long-horizon mode of the path kernel, parent tables in global memory) as
    a FLEET: ltpl_fleet_* against ltpl_planner_* on the same handle, closed loop."""
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c5_lattice
    lat = c5_lattice(horizon=300.0)
    hip = _capi.HipBackend(lat)
    for seed in (2, 3):
        A, B = Planner(hip, 1), Fleet(hip, 3)
        st = drive(lat, A, B, seed, 60, exact=False, scen_b=2)
        assert st['ticks'] >= 40, st
        A.close(); B.close()
