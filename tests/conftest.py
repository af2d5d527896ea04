import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The GPU suite's batches of 64 .. 256 scenarios are there to exercise the batch PIPELINE (one-wave path kernel + lane kernels: what the
# headline runs on). Since round 6 the library serves such small batches with the fused tick kernel by default (include/ltpl_hip.h,
# LTPL_PIPELINE_MIN_SCEN: measured faster below ~2 workgroups per compute unit); the suite pins the rounds-1..5 threshold so that its
# coverage stays what it was, and tests/test_gpu_configs.py::test_small_batches_fused_by_default_match_the_pipeline covers the default.
os.environ.setdefault("LTPL_PIPELINE_MIN_SCEN", "64")
# ... and the form of the velocity stage the headline's 32 768-scenario batches run (follow jobs finished by the lane kernel: default from 8 192
# scenarios on) is engaged for the suite's batches too; test_gpu_vel.py::test_follow_jobs_finished_by_the_lane_kernel_or_by_the_final_kernel
# compares it with the other form bit for bit.
os.environ.setdefault("LTPL_FOLLOW_EMIT_MIN_SCEN", "256")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def monteblanco():
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    return Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))


@pytest.fixture(scope="session")
def oracle_backend(monteblanco):
    from oracle.oracle_lib import OracleBackend
    return OracleBackend(monteblanco)


@pytest.fixture(scope="session")
def hip_backend(monteblanco):
    """The product backend. No fallback: a missing library or device is an error, not a skip."""
    from graphbasedlocaltrajectoryplanner_amd._capi import HipBackend
    return HipBackend(monteblanco)


@pytest.fixture(scope="session")
def open_lattice():
    """Open (unclosed) track: rows 40..339 of the Monteblanco race line, built by the unmodified reference (gen_golden open)."""
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    return Lattice.load(os.path.join(ROOT, "tests", "golden", "open_lattice.npz"))


@pytest.fixture(scope="session")
def oracle_open(open_lattice):
    from oracle.oracle_lib import OracleBackend
    return OracleBackend(open_lattice)


@pytest.fixture(scope="session")
def hip_open(open_lattice):
    from graphbasedlocaltrajectoryplanner_amd._capi import HipBackend
    return HipBackend(open_lattice)
