"""
GPU: the planner entry points of libltpl_hip.so (ltpl_planner_*, ABI v3+) -- the reference's OnlineTrajectoryHandler state
machine in C++ on top of the HIP kernels -- replayed in closed loop against the tick-level recordings of the unmodified
reference (tests/golden/*_ticks.npz): start node, node lists and cut indices bit-exact on every tick, stitched paths, spline
coefficients and trajectories [s, x, y, psi, kappa, vx, ax] within 1e-5 relative (tests/planner_replay.py).
"""
import numpy as np
import pytest

import planner_replay as pr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c2", "c1", "zonewall", "ggdrop", "overtake", "ggmap", "ggmapdrop", "car2ggmap", "car2ggdrop"])
def test_planner_closed_loop_matches_reference_recordings(hip_backend, monteblanco, name):
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    ticks = pr.load_ticks(name)
    planner = Planner(hip_backend, 1)
    seen = pr.replay(planner, monteblanco, ticks)
    assert seen['full'] >= 15
    planner.close()


def test_planner_velocity_smoothing_window(hip_backend, monteblanco):
    """SMOOTHING.filt_window_width = 5 (tph.conv_filt, OTH.py:928-930, :988-990): recording 'filt5' of the unmodified reference."""
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    planner = Planner(hip_backend, 1, filt_window_width=5)
    seen = pr.replay(planner, monteblanco, pr.load_ticks("filt5"))
    assert seen['full'] >= 15
    planner.close()


def test_planner_batch_of_64_matches_single(hip_backend, monteblanco):
    """64 planners in one handle (the batch goes through the one-wave batch kernel) against the recording."""
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    ticks = pr.load_ticks("zonewall")
    planner = Planner(hip_backend, 64)
    pr.replay(planner, monteblanco, ticks, scen=63, n_ticks=200)
    a, b = planner.trajectories(0), planner.trajectories(63)
    for k in a[0]:
        assert np.array_equal(a[0][k][0], b[0][k][0])
    planner.close()


def test_planner_closed_loop_on_an_open_track(hip_open, open_lattice):
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    ticks = pr.load_ticks("open")
    planner = Planner(hip_open, 1)
    seen = pr.replay(planner, open_lattice, ticks)
    assert {"straight", "follow", "right"} <= seen['keys']
    planner.close()
