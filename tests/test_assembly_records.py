"""CPU: the path assembly records ltpl_create derives per node / per edge (csrc/assembly_records.hpp, exported by ltpl_assembly_records) hold
exactly what the path assembly used to read hop by hop (main_online_path_gen.py:260-328 over the arrays of the lattice): the node's first
in-edge and the sources of its first 12 in-edges; the edge's sample range, length, first / last sample and the (sin, cos) of their headings.
Checked on every lattice fixture and on the synthetic ovals; plus the look-up the kernel does on a node record (zero-byte search of the
source among the 12 bytes, paths_team.hpp `team_assemble_rest`) for every edge of the lattice."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSRC, EREC = 12, 10


def records_of(lat):
    import __graft_entry__ as ge
    lib = C.CDLL(ge.build_hip())
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.ltpl_assembly_records.argtypes = [C.c_int32, C.c_int32, pi, pi, pd, pi, pd, pd, pd, pi, pd]
    arr = lambda a, t: np.ascontiguousarray(a, dtype=t)
    in_ptr, src = arr(lat.in_ptr, np.int32), arr(lat.edge_src, np.int32)
    elen, sptr = arr(lat.edge_len, np.float64), arr(lat.samp_ptr, np.int32)
    sx, sy, sp = arr(lat.samples[:, 0], np.float64), arr(lat.samples[:, 1], np.float64), arr(lat.samples[:, 2], np.float64)
    V, E = len(in_ptr) - 1, len(src)
    nrec, erec = np.zeros((V, 4), np.int32), np.zeros((E, EREC), np.float64)
    rc = lib.ltpl_assembly_records(V, E, in_ptr.ctypes.data_as(pi), src.ctypes.data_as(pi), elen.ctypes.data_as(pd), sptr.ctypes.data_as(pi),
                                   sx.ctypes.data_as(pd), sy.ctypes.data_as(pd), sp.ctypes.data_as(pd), nrec.ctypes.data_as(pi), erec.ctypes.data_as(pd))
    assert rc == 0
    return in_ptr, src, elen, sptr, sx, sy, sp, nrec, erec


def check(lat):
    in_ptr, src, elen, sptr, sx, sy, sp, nrec, erec = records_of(lat)
    V, E = len(in_ptr) - 1, len(src)
    # node records
    assert np.array_equal(nrec[:, 0], in_ptr[:-1])
    b = nrec[:, 1:].copy().view(np.uint8).reshape(V, NSRC)
    deg = in_ptr[1:] - in_ptr[:-1]
    for k in range(NSRC):
        has = deg > k
        assert np.array_equal(b[has, k], src[in_ptr[:-1][has] + k].astype(np.uint8))
        assert np.all(b[~has, k] == 0xff)
    assert int(src.max()) < 0x80                                   # the byte search below relies on it (one-byte parents: <= 127 nodes per layer)
    # the kernel's look-up: first byte equal to the source, longer in-edge lists continue in edge_src
    dst = np.repeat(np.arange(V), deg)
    k_true = np.arange(E) - in_ptr[:-1][dst]
    first = np.full(E, NSRC, np.int64)
    eq = b[dst] == src[:, None].astype(np.uint8)
    anyeq = eq.any(axis=1)
    first[anyeq] = eq[anyeq].argmax(axis=1)
    short = k_true < NSRC
    assert np.array_equal(first[short], k_true[short])             # found among the 12 bytes at its own position (sources of a node are distinct)
    assert np.all(first[~short] == NSRC)                           # beyond: not among them -> the kernel scans edge_src from position 12 on
    # edge records
    w = erec[:, 0].copy().view(np.uint64)
    k0, ns = (w & 0xffffffff).astype(np.int64), (w >> 32).astype(np.int64)
    assert np.array_equal(k0, sptr[:-1]) and np.array_equal(ns, sptr[1:] - sptr[:-1]) and int(ns.min()) >= 1
    assert np.array_equal(erec[:, 1], elen)
    k1 = sptr[1:] - 1
    assert np.array_equal(erec[:, 2], sx[k0]) and np.array_equal(erec[:, 3], sy[k0])
    assert np.array_equal(erec[:, 4], sx[k1]) and np.array_equal(erec[:, 5], sy[k1])
    # (libm's sin / cos on the host against numpy's: the same routine on this platform; one ulp is the promise)
    for col, ref in ((6, np.sin(sp[k0])), (7, np.cos(sp[k0])), (8, np.sin(sp[k1])), (9, np.cos(sp[k1]))):
        assert float(np.abs(erec[:, col] - ref).max()) <= 2.3e-16


@pytest.mark.parametrize("fixture", sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "tests", "golden", "*_lattice.npz"))))
def test_records_match_the_lattice_arrays(fixture):
    check(Lattice.load(os.path.join(ROOT, "tests", "golden", fixture)))


def test_records_of_the_synthetic_ovals():
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice, c5_lattice
    check(c3_lattice())
    check(c5_lattice(horizon=100.0))
