"""
SURVEY.md section 8f rank 3: the offline lattice build (race line file + offline parameters -> struct-of-arrays lattice) against the
lattices exported from the REFERENCE's own GraphBase after its offline build (tests/golden/monteblanco_lattice.npz: closed track;
open_lattice.npz: unclosed track; both written by oracle/gen_golden.py through Lattice.from_graph_base).
Topology (nodes per layer, race line indices, CSC pointers, edge sources, sample pointers) must be IDENTICAL, every float column
within 1e-5 relative (observed: 1e-13).

  CPU  host logic of build_lattice (layer selection, node skeleton, candidate edges, pruning, cost, assembly) with the per-edge
       arithmetic taken from oracle/offline_edges_ref.py (reference formulation in NumPy, test infrastructure)
  GPU  the same with the per-edge arithmetic on the device (ltpl_offline_edges, k_offline_edges), plus kernel vs restatement
"""
import os

import numpy as np
import pytest

from helpers import ROOT, assert_close_rel
from graphbasedlocaltrajectoryplanner_amd import offline_build as ob
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice

INT_COLS = ("nodes_in_layer", "raceline_index", "in_ptr", "edge_src", "samp_ptr")
FLOAT_COLS = ("s_raceline", "refline", "raceline", "vel_raceline", "normvec", "track_width_right", "track_width_left", "alpha",
              "node_pos", "node_psi", "vgoal_cost", "edge_cost", "edge_len", "edge_coeff", "glob_rl")


def track(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name + "_track.npz")) as z:
        return {k: z[k] for k in z.files}


def check_against_reference_export(lat, name):
    ref = Lattice.load(os.path.join(ROOT, "tests", "golden", name + "_lattice.npz"))
    assert (lat.num_layers, lat.num_nodes, lat.num_edges, lat.num_samples, lat.closed) == \
        (ref.num_layers, ref.num_nodes, ref.num_edges, ref.num_samples, ref.closed)
    for k in INT_COLS:
        assert np.array_equal(getattr(lat, k), getattr(ref, k)), k                       # bit-exact topology
    for k in FLOAT_COLS:
        assert_close_rel(getattr(lat, k), getattr(ref, k), what=k)
    for col, nm in enumerate(("x", "y", "psi", "kappa", "el")):
        if nm == "psi":
            d = np.abs(np.mod(lat.samples[:, col] - ref.samples[:, col] + np.pi, 2 * np.pi) - np.pi)
            assert float(d.max()) <= 1e-5 * np.pi
        else:
            assert_close_rel(lat.samples[:, col], ref.samples[:, col], what="samples " + nm, floor=1e-3)
    for k in ("lat_resolution", "lat_offset", "veh_width", "veh_length", "sampled_resolution", "vel_decrease_lat",
              "min_plan_horizon", "plan_horizon_mode", "virt_goal_node_cost"):
        assert getattr(lat, k) == getattr(ref, k), k


@pytest.mark.parametrize("name", ["monteblanco", "open"])
def test_host_logic_reproduces_the_reference_lattice(name):
    from oracle import offline_edges_ref
    lat = ob.build_lattice(track(name), ob.OFFLINE_DEFAULTS, offline_edges_ref.evaluate)
    check_against_reference_export(lat, name)


@pytest.mark.parametrize("name", ["berlin", "modena"])
def test_host_logic_reproduces_the_reference_lattice_fingerprint(name):
    """Tracks whose reference-built lattice is not committed in full (5 MB each): sizes, SHA-256 of every topology column and the
    moments of the float columns of the lattice the REFERENCE built (tests/golden/lattice_digests.json, oracle/gen_golden.py digests)."""
    import json
    from oracle import offline_edges_ref
    from oracle.gen_golden import lattice_digest
    with open(os.path.join(ROOT, "tests", "golden", "lattice_digests.json")) as fh:
        ref = json.load(fh)[name]
    got = lattice_digest(ob.build_lattice(track(name), ob.OFFLINE_DEFAULTS, offline_edges_ref.evaluate))
    assert got["sizes"] == ref["sizes"]
    for k in ref:
        if k.startswith("sha_"):
            assert got[k] == ref[k], k                                                   # bit-exact topology
        elif k.startswith("mom_"):
            scale = max(abs(ref[k][1]), 1e-9)                                            # sum of magnitudes
            assert abs(got[k][0] - ref[k][0]) <= 1e-9 * scale and abs(got[k][1] - ref[k][1]) <= 1e-9 * scale, k
            assert abs(got[k][2] - ref[k][2]) <= 1e-9 * max(ref[k][2], 1e-9), k
            assert abs(got[k][3] - ref[k][3]) <= 1e-7 * max(abs(ref[k][3]), 1e-3) and abs(got[k][4] - ref[k][4]) <= 1e-7 * max(abs(ref[k][4]), 1e-3), k


def test_vehicle_too_wide_is_rejected():
    from oracle import offline_edges_ref
    cfg = dict(ob.OFFLINE_DEFAULTS, veh_width=9.0)
    with pytest.raises(ValueError):
        ob.build_lattice(track("monteblanco"), cfg, offline_edges_ref.evaluate)


def test_built_lattice_round_trips_through_the_portable_format(tmp_path):
    from oracle import offline_edges_ref
    lat = ob.build_lattice(track("open"), ob.OFFLINE_DEFAULTS, offline_edges_ref.evaluate)
    p = str(tmp_path / "lattice.npz")
    lat.save(p)
    back = Lattice.load(p)
    for k in INT_COLS + FLOAT_COLS + ("samples",):
        assert np.array_equal(getattr(lat, k), getattr(back, k)), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["monteblanco", "open"])
def test_device_build_reproduces_the_reference_lattice(hip_backend, name):
    lat = ob.build_lattice(track(name), ob.OFFLINE_DEFAULTS, ob.edges_on_device(hip_backend.lib))
    check_against_reference_export(lat, name)


@pytest.mark.gpu
def test_device_edges_match_the_restatement(hip_backend):
    from oracle import offline_edges_ref
    seen = {}

    def both(*args):
        seen["dev"] = ob.edges_on_device(hip_backend.lib)(*args)
        seen["ref"] = offline_edges_ref.evaluate(*args)
        return seen["dev"]
    ob.build_lattice(track("monteblanco"), ob.OFFLINE_DEFAULTS, both)
    d, r = seen["dev"], seen["ref"]
    assert np.array_equal(d["n_samples"], r["n_samples"]) and np.array_equal(d["valid"], r["valid"])
    assert d["n_samples"].size > 30000                       # every candidate edge, not only the survivors
    for k in ("coeff", "length", "kappa_avg", "kappa_range"):
        assert_close_rel(d[k], r[k], what=k)
    assert_close_rel(d["samples"][..., 0:2], r["samples"][..., 0:2], what="xy")
    assert_close_rel(d["samples"][..., 3:5], r["samples"][..., 3:5], what="kappa, el")


@pytest.mark.gpu
def test_planner_runs_on_a_device_built_lattice(hip_backend):
    """End to end: build on the device, upload, plan -- the built lattice is what ltpl_create takes."""
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from scenarios import random_scenarios
    from oracle.oracle_lib import OracleBackend
    lat = ob.build_lattice(track("monteblanco"), ob.OFFLINE_DEFAULTS, ob.edges_on_device(hip_backend.lib))
    hip2 = _capi.HipBackend(lat)
    scen, _ = random_scenarios(lat, 80, seed=9)
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    res, ref = hip2.plan_paths(batch), OracleBackend(lat).plan_paths(batch)
    assert np.array_equal(res.valid, ref.valid) and np.array_equal(res.n_nodes * res.valid, ref.n_nodes * ref.valid)
    hip2.close()
