"""GPU: the cross-lane helpers of the kernels (DPP row operations + v_permlane16/32_swap, csrc/ltpl_hip.hip "CROSS-LANE MOVES ON THE
VECTOR ALU") against plain loops over the wave's values -- wave minima / maxima, the lexicographic two- and three-key minima with
many exact ties, the segment merge of the closest-layer search for every segment width, the exclusive prefix sum.
The check kernel lives in the experiment build (libltpl_hip_exp.so: same source, same helpers)."""
import ctypes as C

import numpy as np
import pytest

from graphbasedlocaltrajectoryplanner_amd import _capi

pytestmark = pytest.mark.gpu


def test_cross_lane_helpers_match_plain_loops():
    lib = C.CDLL(_capi.experiment_library_path())
    assert hasattr(lib, "ltpl_exp_wave_ops_check")
    rng = np.random.default_rng(5)
    rounds = 512
    vals = np.empty((rounds, 128), np.float64)
    ivals = np.empty((rounds, 64), np.int32)
    for r in range(rounds):
        kind = r % 4
        if kind == 0:      # distinct keys
            vals[r] = rng.random(128) * 1e3
            ivals[r] = rng.permutation(1 << 16)[:64]
        elif kind == 1:    # many exact ties on the first key, some on the second
            vals[r, :64] = rng.integers(0, 4, 64)
            vals[r, 64:] = rng.integers(0, 3, 64)
            ivals[r] = rng.integers(0, 1 << 20, 64)
        elif kind == 2:    # +inf among the keys (unreachable nodes), all keys equal in some rounds
            vals[r] = np.where(rng.random(128) < 0.5, np.inf, rng.integers(0, 2, 128).astype(np.float64))
            if r % 8 == 2:
                vals[r] = np.inf
            ivals[r] = rng.integers(-(1 << 30), 1 << 30, 64)
        else:              # the minimum in one given lane
            vals[r] = 5.0 + rng.random(128)
            vals[r, r % 64] = 1.0
            ivals[r] = np.arange(64)[::-1]
    n_bad = C.c_int32(-1)
    rc = lib.ltpl_exp_wave_ops_check(0, vals.ctypes.data_as(C.POINTER(C.c_double)), ivals.ctypes.data_as(C.POINTER(C.c_int32)),
                                     rounds, C.byref(n_bad))
    assert rc == 0
    assert n_bad.value == 0


def test_heading_atan2_matches_numpy():
    """heading_atan2 (csrc/paths_team.hpp: the heading of a re-sampled path point, octant reduction + series instead of the library
    routine) against numpy.arctan2: all octants, the reduction's break points, the axes, magnitudes from 1e-6 to 1e6."""
    lib = C.CDLL(_capi.experiment_library_path())
    assert hasattr(lib, "ltpl_exp_heading_atan2")
    rng = np.random.default_rng(9)
    ang = np.concatenate((rng.uniform(-np.pi, np.pi, 200000), np.arange(-16, 17) * np.pi / 16, np.arange(-16, 17) * np.pi / 16 + 1e-9,
                          np.arange(-16, 17) * np.pi / 16 - 1e-9))
    mag = 10.0 ** rng.uniform(-6, 6, ang.size)
    y, x = mag * np.sin(ang), mag * np.cos(ang)
    y = np.concatenate((y, [0.0, 0.0, 1.0, -1.0, 3.0, -3.0, 0.0])); x = np.concatenate((x, [1.0, -1.0, 0.0, 0.0, 3.0, -3.0, 2.5]))
    out = np.empty_like(y)
    pd = C.POINTER(C.c_double)
    rc = lib.ltpl_exp_heading_atan2(0, y.ctypes.data_as(pd), x.ctypes.data_as(pd), out.ctypes.data_as(pd), y.size)
    assert rc == 0
    ref = np.arctan2(y, x)
    d = np.abs(out - ref)
    d = np.minimum(d, 2 * np.pi - d)                  # (+pi and -pi are the same heading)
    assert float(d.max()) <= 4e-15, float(d.max())


def test_heading_sincos_matches_numpy():
    """heading_sincos (csrc/paths_team.hpp: sin / cos of a scenario's start heading) against numpy on [-12, 12] (the kernel calls the library
    routine beyond), the quadrant borders and zero."""
    lib = C.CDLL(_capi.experiment_library_path())
    assert hasattr(lib, "ltpl_exp_heading_sincos")
    rng = np.random.default_rng(10)
    x = np.concatenate((rng.uniform(-12.0, 12.0, 200000), np.arange(-15, 16) * np.pi / 4, np.arange(-15, 16) * np.pi / 4 + 1e-12, [0.0, 1e-300, -1e-9]))
    sn, cs = np.empty_like(x), np.empty_like(x)
    pd = C.POINTER(C.c_double)
    assert lib.ltpl_exp_heading_sincos(0, x.ctypes.data_as(pd), sn.ctypes.data_as(pd), cs.ctypes.data_as(pd), x.size) == 0
    assert float(np.abs(sn - np.sin(x)).max()) <= 4e-16 and float(np.abs(cs - np.cos(x)).max()) <= 4e-16
