"""CPU: the C-ABI library loads (no GPU needed) and exports every symbol include/ltpl_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ltpl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ltpl_[a-z_]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("ltpl_create", "ltpl_destroy", "ltpl_plan_paths", "ltpl_vel_profile", "ltpl_tick_batch",
                 "ltpl_last_error", "ltpl_version", "ltpl_get_caps", "ltpl_batch_upload", "ltpl_batch_run",
                 "ltpl_batch_download", "ltpl_planner_create", "ltpl_planner_calc_paths", "ltpl_planner_calc_vel_profile",
                 "ltpl_planner_set_start", "ltpl_planner_get_paths", "ltpl_planner_get_trajectories"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    lib_path = ge.build_hip()
    lib = ctypes.CDLL(lib_path)
    for sym in declared_symbols():
        assert hasattr(lib, sym), "libltpl_hip.so does not export %s" % sym
    assert lib.ltpl_version() == 9          # v6: per-planner vel_max / machine tables in the fleet, ltpl_fleet_set_start_range;
                                            # v7 (additive): ltpl_paths_kernel_symbol, ltpl_layer_grid, ltpl_fleet_digest;
                                            # v8 (additive): ltpl_assembly_records; v9 (additive): ltpl_create_ex, ltpl_tick_persistent_*


def test_product_fails_loudly_without_library(monteblanco, tmp_path):
    from graphbasedlocaltrajectoryplanner_amd import _capi
    with pytest.raises(_capi.BackendError):
        _capi.HipBackend(monteblanco, lib_path=str(tmp_path / "missing.so"))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "graphbasedlocaltrajectoryplanner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").replace("the oracle", "") or f == "__init__.py", \
                    "%s mentions the oracle" % f


def test_path_kernels_have_no_register_spills():
    """__graft_entry__.build() rejects a library whose path kernels exceed their register budget (see check_no_register_spills)."""
    import __graft_entry__ as g
    g.build_hip()
    res = {k: v for k, v in g.kernel_resources(g.HIP_LIB).items() if "k_tick_persistent" not in k}     # (one resident workgroup: own rule, no scratch)
    paths = {k: v for k, v in res.items() if "k_paths" in k or "k_tick" in k}
    assert len(paths) >= 3                                  # runtime plan (1 and 4 waves) + compile-time plan classes
    for name, r in paths.items():
        # at most a few values parked in scratch across the sweeps; check_no_register_spills verifies on the ISA that no scratch
        # instruction sits in a layer loop of the sweeps, and that the fixed-plan batch kernels keep 4 waves per SIMD
        assert r["vgpr_spill_count"] <= g.parked_limit(name) <= g.PARKED_VGPRS_MAX and r["private_segment_fixed_size"] <= g.PARKED_BYTES_MAX, name
        if "k_pathsILi1E6PlanFx" not in name:
            assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, name
        if "k_pathsILi1E6PlanFx" in name:
            assert r["vgpr_count"] <= 128, name
    assert all(r["vgpr_spill_count"] == 0 for k, r in res.items() if k not in paths and "k_vel_lanes" not in k)
    # k_vel_lanes is held to 256 registers (two waves per SIMD: what it displaces next to a resident path kernel) and parks a few values for it
    assert all(r["vgpr_count"] <= 256 and r["vgpr_spill_count"] <= (48 if "k_vel_lanesILi0E" in k else 8) for k, r in res.items() if "k_vel_lanes" in k)
    g.check_no_register_spills(g.HIP_LIB)
    loops = g.layer_loop_ops(g.HIP_LIB, [k for k in paths if "k_pathsILi1E6PlanFx" in k])
    assert loops and all(m["loops"] >= 1 and m["scratch"] == 0 for m in loops.values()), loops


def test_create_ex_and_the_persistent_entry_points_check_their_arguments_without_a_device():
    """ABI v9 (round 6): argument errors of ltpl_create_ex / ltpl_tick_persistent_* come back as status codes before any HIP call (no GPU here)."""
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_hip())
    lib.ltpl_create_ex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
    lib.ltpl_last_error.restype = ctypes.c_char_p
    lib.ltpl_last_error.argtypes = [ctypes.c_void_p]
    h = ctypes.c_void_p()
    desc = (ctypes.c_char * 2048)()                                           # (never read: the flag check comes first)
    assert lib.ltpl_create_ex(ctypes.byref(desc), 0, 0x80, ctypes.byref(h)) == 1 and b"unknown flag" in lib.ltpl_last_error(None)
    assert lib.ltpl_create_ex(None, 0, 1, ctypes.byref(h)) == 1                # LTPL_ERR_INVALID_ARG: null lattice
    lib.ltpl_tick_persistent_stop.argtypes = [ctypes.c_void_p]
    lib.ltpl_tick_persistent_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert lib.ltpl_tick_persistent_stop(None) == 1 and lib.ltpl_tick_persistent_stats(None, None) == 1
