#!/usr/bin/env python
"""
bench.py -- planning ticks/s of the MI355X tick pipeline (seam 1 + per-primitive velocity stage) on the C2 workload.

  python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: the script spawns N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the tick pipeline over one batch of independent C2 scenarios (Monteblanco lattice, 4 action
primitives, 8 dynamic opponents with a 0.2 s prediction each = 16 obstacle positions, sample zone) whose inputs are already
resident in HBM (ltpl_batch_upload). Every rank owns one GPU and its own shard of scenarios (weak scaling, no data-path
collective: scenarios are independent); torch.distributed (RCCL) is only used for the barrier and the
max-over-ranks of the timed region. The timed region is the K-step block repeated until it lasts >= 2 s (`timed_steps` in the
line; `ms_per_step` and `value` are per step), bracketed by barrier + synchronize on both sides.

The JSON line also carries
  roofline        ALGORITHMIC bytes per launch (graphbasedlocaltrajectoryplanner_amd/roofline.py) of the
                  dominant kernel (path kernel: mask + sweeps + spline) divided by its average duration measured with HIP
                  events on the library's own stream inside the timed region, against the 8 TB/s HBM peak; `traffic` = HBM
                  bytes per launch from the rocprofv3 PMC passes (profiles/pmc_traffic.json) -- the working set is cache
                  resident, so the kernel is issue / latency bound, not DRAM bound
  cpu_baseline    the oracle's plain-C restatement (kind "port", 1 core) timed on a bounded sample of the same scenarios
  parity_checked  the GPU results of those sample scenarios compared with the oracle's results of the cpu_baseline leg
  latency_us      single-scenario ticks: `p50/p99` one synchronous ltpl_tick_batch call including packing, PCIe and unpacking
                  into the reference's Python structures; `dropin_*` one full closed-loop tick of the planner entry points
                  (ltpl_planner_calc_paths + ltpl_planner_calc_vel_profile + copy-out) replaying the recorded C2 loop;
                  `persistent_tick` the same single ticks served by the RESIDENT kernel (ltpl_create_ex(LTPL_CREATE_PERSISTENT_TICK):
                  no launch, no copy call; own handle, outputs compared bit for bit with the launched kernel's)
  extra           three_slot_ticks_per_s (every scenario has an opponent 20-80 m ahead: three primitives live),
                  pcie_inclusive (host buffers in and out per call), closed_loop* (host planner / device-resident fleet),
                  c4 (BASELINE config C4 as stated: 1024 scenarios in one call and the 128-scenario shard of one of 8 GPUs, resident and
                  PCIe-inclusive), c3 (the "HBM roofline run" with its own roofline / binding / parity), c5 (high-resolution latency run)

  --scaling strong --batch-total T   a FIXED batch of T scenarios block-partitioned over the ranks (C4 as a bench mode) instead of the
                  default weak scaling (--batch scenarios per GPU); the line then says "scaling": "strong" and `value` counts T per step.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from graphbasedlocaltrajectoryplanner_amd import _capi                       # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice             # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.roofline import algorithmic_bytes  # noqa: E402
from graphbasedlocaltrajectoryplanner_amd.scenario_gen import c2_scenarios   # noqa: E402

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MIN_TIMED_S = 2.0
TARGET_TICKS_PER_S = 10000.0    # BASELINE.json north_star / BASELINE.md section 3: C2 target on one MI355X
W_LAST = [0.0, 0.5, 0.8]        # params/ltpl_config_online.ini:71


def make_batch(lat, n, seed, workload="c2"):
    if workload == "c3":
        from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import scattered_obstacle_scenarios
        scen, vels = scattered_obstacle_scenarios(lat, n, n_obj=32, seed=seed)
    elif workload == "c2_near":
        scen, vels = c2_scenarios(lat, n, seed=seed, lead_gap=(20.0, 80.0))
    else:
        scen, vels = c2_scenarios(lat, n, seed=seed)
    rng = np.random.default_rng(seed + 77)
    params = _capi.VelParamSet(len_veh=lat.veh_length)      # Graph_LTPL.calc_vel_profile defaults (Graph_LTPL.py:347-351)
    vplan = rng.uniform(5.0, 60.0, n)
    pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    batch = _capi.PathsBatch(scen, w_last_edges=W_LAST)
    vel = _capi.TickVelBatch(params, n, vplan, vplan, pos, np.concatenate(vels))
    return scen, batch, vel


def sample_indices(n_batch, n_sample):
    """Scenario indices of the CPU / parity sample: spread evenly over the WHOLE batch (first and last scenario included)."""
    n_sample = max(1, min(n_sample, n_batch))
    return np.unique(np.linspace(0, n_batch - 1, n_sample).round().astype(np.int64))


def cpu_baseline(lat, scen_batch, batch, vel, idx):
    """Oracle (plain-C restatement, oracle/ltpl_oracle.c) on the scenarios ``idx`` of rank 0's shard (a strided sample of the whole
    batch), 1 core. Returns the JSON object and the oracle's results (used as the checker of `parity_checked`)."""
    from oracle.oracle_lib import OracleBackend
    orc = OracleBackend(lat)
    scen = [scen_batch[int(i)] for i in idx]
    sub = _capi.PathsBatch(scen, w_last_edges=W_LAST)
    vo = np.asarray(batch.veh_off)
    veh_vel = np.concatenate([vel.veh_vel[vo[int(i)]:vo[int(i) + 1]] for i in idx]) if len(idx) else np.zeros(0)
    v = _capi.TickVelBatch(vel.params, len(scen), vel.vel_plan[idx], vel.vel_est[idx],
                           np.column_stack((vel.pos_x[idx], vel.pos_y[idx])), veh_vel)
    ref = orc.tick_batch(sub, v)                                # warm caches; kept as the parity reference
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.tick_batch(sub, v)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 200:
            break
    out = {"value": len(scen) * reps / el, "unit": "ticks/s", "cores": 1, "kind": "port",
           "sample": "%d scenarios spread evenly over the batch of the timed region x %d passes through oracle_tick_batch (plain C, "
                     "-O2, single thread, timed in this run on the GPU box's host). The reference's own Python cannot run here "
                     "(/root/reference is absent on the GPU box): its measured rate is the committed artefact cited in `reference_python`"
                     % (len(scen), reps)}
    # the unmodified reference over the dependency shims, C2 closed loop, timed in the build container by tools/ref_python_rate.py
    # (BASELINE.md section 4 / SURVEY.md section 8d: >= 2 000 ticks after 100 warm-up, perf_counter around calc_paths + calc_vel_profile)
    p = os.path.join(ROOT, "profiles", "r05_ref_python_cpu.json")
    if os.path.isfile(p):
        try:
            with open(p) as fh:
                d = json.load(fh)
            out["reference_python"] = {"ticks_per_s": d["ticks_per_s"], "mean_ms": d["tick"]["mean_ms"], "p50_ms": d["tick"]["p50_ms"],
                                       "p99_ms": d["tick"]["p99_ms"], "ticks": d["ticks"], "warmup_ticks": d["warmup_ticks"], "cores": d["cores"],
                                       "cpu_model": d["cpu_model"], "label": "reference-Python-over-shim", "where": d["where"],
                                       "source": "profiles/r05_ref_python_cpu.json (tools/ref_python_rate.py); not measured in this run"}
        except Exception:
            pass
    return out, ref


ELEM_TOL_VX = 1e-5      # element-wise bounds, sample by sample: vx where |vx| >= 1 m/s, ax where |ax| >= 0.5 m/s^2 (north_star: "velocity profiles
ELEM_TOL_AX = 1e-5      # within 1e-5 relative"; the array-level bounds are 1e-5 of the array's largest value)


def parity_check(res, vres, ref, idx):
    """GPU results of the scenarios ``idx`` of the TIMED batch against the oracle's results of the same scenarios: every integer output
    bit-exact, every float quantity within 1e-5 relative against ITS OWN scale (tests/helpers.py) -- reported per quantity. The oracle's
    fused tick is itself pinned to the unmodified reference: tests/test_fresh_tick_golden.py."""
    ores, ovres = ref
    ints = {k: 0 for k in ("n_actions", "closest_obj_index", "closest_obj_node", "end_layer", "action_id", "valid", "reduced", "goal_layer",
                           "n_nodes", "n_pts", "n_ties", "nodes", "node_idx", "vel_bound", "too_close", "el_length_column")}
    worst = {k: 0.0 for k in ("x", "y", "psi", "kappa", "coeff_a0", "coeff_a1", "coeff_a2", "coeff_a3", "vx", "ax")}
    n_paths = 0
    # ELEMENT-WISE relative errors |a - d| / |d| over the samples whose reference magnitude lies above a floor (the per-array max-norm
    # tolerances above are the assertions of the parity suite; these show what they allow element by element, e.g. on the slow tail of a
    # profile that brakes to standstill): vx where |vx| >= 1 m/s, ax where |ax| >= 0.5 m/s^2, kappa where |kappa| >= 1e-3 1/m
    ELEM_FLOOR = {"vx": 1.0, "ax": 0.5, "kappa": 1e-3}
    elem = {k: [] for k in ELEM_FLOOR}

    def rel(name, a, b, scale):
        worst[name] = max(worst[name], float(np.max(np.abs(a - b))) / scale if a.size else 0.0)
        if name in elem and a.size:
            m = np.abs(b) >= ELEM_FLOOR[name]
            if m.any():
                elem[name].append(np.abs(a[m] - b[m]) / np.abs(b[m]))

    for k, s in enumerate(idx):
        s = int(s)
        for name in ("n_actions", "closest_obj_index", "end_layer"):
            ints[name] += int(getattr(res, name)[s] != getattr(ores, name)[k])
        ints["closest_obj_node"] += int(not np.array_equal(res.closest_obj_node[s], ores.closest_obj_node[k]))
        na = int(ores.n_actions[k])
        for name in ("action_id", "valid", "reduced", "goal_layer"):
            ints[name] += int(not np.array_equal(getattr(res, name)[s, :na], getattr(ores, name)[k, :na]))
        for a in range(na):
            if not ores.valid[k, a] or not res.valid[s, a]:
                continue
            n_paths += 1
            nn, npts = int(ores.n_nodes[k, a]), int(ores.n_pts[k, a])
            for name in ("n_nodes", "n_pts", "n_ties"):
                ints[name] += int(getattr(res, name)[s, a] != getattr(ores, name)[k, a])
            ints["vel_bound"] += int(vres.vel_bound[s, a] != ovres.vel_bound[k, a])
            ints["too_close"] += int(vres.too_close[s, a] != ovres.too_close[k, a])
            if int(res.n_nodes[s, a]) != nn or int(res.n_pts[s, a]) != npts or not np.array_equal(res.nodes[s, a, :nn], ores.nodes[k, a, :nn]):
                ints["nodes"] += 1
                continue
            ints["node_idx"] += int(not np.array_equal(res.node_idx[s, a, :nn], ores.node_idx[k, a, :nn]))
            pp, opp = res.path_param[s, a, :npts], ores.path_param[k, a, :npts]
            ints["el_length_column"] += int(not np.array_equal(pp[:, 4], opp[:, 4]))           # a copy of the lattice's numbers: bit-exact
            rel("x", pp[:, 0], opp[:, 0], max(float(np.ptp(opp[:, 0])), 1.0))                  # against the extent of the path
            rel("y", pp[:, 1], opp[:, 1], max(float(np.ptp(opp[:, 1])), 1.0))
            dpsi = np.abs(np.mod(pp[:, 2] - opp[:, 2] + np.pi, 2 * np.pi) - np.pi)
            worst["psi"] = max(worst["psi"], float(dpsi.max()) / np.pi)
            rel("kappa", pp[:, 3], opp[:, 3], max(float(np.max(np.abs(opp[:, 3]))), 1e-4))   # floor 1e-4 1/m
            co, oco = res.coeff[s, a, :nn - 1], ores.coeff[k, a, :nn - 1]
            for c in (0, 4):
                rel("coeff_a0", co[:, c], oco[:, c], max(float(np.ptp(oco[:, c])), 1.0))
            for o in (1, 2, 3):
                rel("coeff_a%d" % o, co[:, [o, 4 + o]], oco[:, [o, 4 + o]], max(float(np.max(np.abs(oco[:, [o, 4 + o]]))), 1e-3))
            vx, ovx = vres.vx[s, a, :npts], ovres.vx[k, a, :npts]
            rel("vx", vx, ovx, max(float(np.max(np.abs(ovx))), 1.0))                           # floor 1 m/s
            rel("ax", vres.ax[s, a, :npts], ovres.ax[k, a, :npts], max(float(np.max(np.abs(ovx))) ** 2 / 2.0, 5.0))   # ax differentiates v^2
    bad_ints = {k: v for k, v in ints.items() if v}
    max_rel = max(worst.values()) if worst else 0.0
    elementwise = {}
    for k, parts in elem.items():
        e = np.concatenate(parts) if parts else np.zeros(0)
        elementwise[k] = {"samples": int(e.size), "floor": ELEM_FLOOR[k],
                          "p50": float(np.percentile(e, 50)) if e.size else None, "p99": float(np.percentile(e, 99)) if e.size else None,
                          "max": float(e.max()) if e.size else None}
    # north_star: "velocity profiles within 1e-5 relative" is asserted per array against the array's scale; element by element the
    # velocity must stay within ELEM_TOL_VX wherever the car moves at all (|vx| >= 1 m/s)
    vx_elem_ok = elementwise["vx"]["max"] is None or elementwise["vx"]["max"] <= ELEM_TOL_VX
    ax_elem_ok = elementwise["ax"]["max"] is None or elementwise["ax"]["max"] <= ELEM_TOL_AX
    ok = not bad_ints and max_rel <= 1e-5 and vx_elem_ok and ax_elem_ok
    return ok, {"scenarios": int(len(idx)), "paths": n_paths, "sample": "evenly spread over the %d scenarios of the timed batch" % res.n_scen,
                "max_rel_err": max_rel, "max_rel_err_by_quantity": worst,
                "elementwise_rel_err": dict(elementwise, what="|gpu - oracle| / |oracle| per sample above the floor (vx >= 1 m/s, ax >= 0.5 m/s^2, "
                                                              "kappa >= 1e-3 1/m): p50 / p99 / max; asserted: max(vx) <= %.0e, max(ax) <= %.0e" % (ELEM_TOL_VX, ELEM_TOL_AX)),
                "integer_outputs_compared_bit_exact": sorted(ints.keys()), "integer_mismatches": bad_ints,
                "scales": "x, y, coeff_a0: extent of the path; coeff_a1..a3: largest magnitude of that order; kappa: floor 1e-4 1/m; "
                          "psi: pi; vx: floor 1 m/s; ax: max(v^2 / 2, 5 m/s^2)"}


def library_stamp(hip):
    """Resolved path + hashes of the library this run drives (LTPL_HIP_LIB may redirect it) and the digest of the batch path kernel's
    instruction stream (__graft_entry__.build_stamp): what the committed counter passes under profiles/ are matched against."""
    import __graft_entry__ as ge
    sym = hip.paths_kernel_symbol(1)
    try:
        st = ge.build_stamp(hip.lib_path, sym) if sym else {"lib_sha256": ge.file_sha256(hip.lib_path)}
    except Exception as e:                                   # (llvm-objdump missing, ...: reported, not fatal)
        st = {"error": "%s: %s" % (type(e).__name__, e)}
    st["path"] = hip.lib_path
    return st


PROFILES_DIR = None      # (tests point this at a scratch directory)


def _read_pmc(name, batch, workload, stamp):
    """A committed rocprofv3 PMC summary profiles/<name> if it was collected for this workload and batch size (grid = 64 threads x batch);
    `build_matches`: the digest of the profiled library's path kernel (written by tools/pmc_ab.sh / tools/profile_summarise.py) equals
    the one of the library this run drives -- false flags a stale file, null a file without a stamp (collected before round 5)."""
    p = os.path.join(PROFILES_DIR or os.path.join(ROOT, "profiles"), name)
    if not os.path.isfile(p):
        return None
    try:
        with open(p) as fh:
            d = json.load(fh)
    except Exception:
        return None
    if d.get("workload", "c2") != workload or int(d.get("grid_size", 0)) != 64 * batch:
        return None
    prof = (d.get("build") or {}).get("isa_sha256")
    mine = (stamp or {}).get("isa_sha256")
    d["build_matches"] = (prof == mine) if (prof and mine) else None
    return d


def read_traffic(batch, workload, stamp=None):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary (profiles/pmc_traffic.json, C3:
    pmc_traffic_c3.json), or None."""
    return _read_pmc("pmc_traffic.json" if workload == "c2" else "pmc_traffic_%s.json" % workload, batch, workload, stamp)


def read_issue(batch, workload, stamp=None):
    """VALU / SALU / LDS instruction counts, lane utilisation and wait cycles of the dominant kernel from a committed rocprofv3 PMC pass
    (profiles/pmc_issue.json, C3: pmc_issue_c3.json; written by tools/pmc_ab.sh), or None."""
    return _read_pmc("pmc_issue.json" if workload == "c2" else "pmc_issue_%s.json" % workload, batch, workload, stamp)


def issue_summary(issue_pmc, batch, dom_ms):
    """The binding-pipe figures of a counter pass (see the `binding` block of the line) at the kernel's live duration dom_ms."""
    if not issue_pmc:
        return None
    simd_cycles = N_SIMD * dom_ms * 1e-3 * CLOCK_HZ
    return {"valu_insts": issue_pmc["valu_insts_per_launch"], "lanes_active": issue_pmc["lanes_active_per_valu_inst"],
            # SQ_ACTIVE_INST_VALU (quad-cycles with a VALU instruction in flight, summed over the SIMDs) over the SIMD-cycles of the
            # launch at its live duration; without that counter: instructions x 4 cycles (fp64 / transcendental rate; fp32 and
            # integer wave64 instructions issue in 2 on CDNA4's SIMD-32, so this is an upper bound then)
            "valu_util": (issue_pmc.get("valu_active_quad_cycles_per_launch") or issue_pmc["valu_insts_per_launch"])
            * VALU_CYCLES_PER_INST / simd_cycles,
            "salu_insts": issue_pmc.get("salu_insts_per_launch"), "lds_insts": issue_pmc.get("lds_insts_per_launch"),
            "valu_insts_per_scenario": issue_pmc["valu_insts_per_launch"] / batch,
            "salu_insts_per_scenario": (issue_pmc.get("salu_insts_per_launch") or 0.0) / batch,
            "lds_insts_per_scenario": (issue_pmc.get("lds_insts_per_launch") or 0.0) / batch,
            # the LDS pipe: one per CU, shared by its 16 resident scenarios (SQ_ACTIVE_INST_LDS, quad-cycles)
            "lds_util": (issue_pmc["lds_active_quad_cycles_per_launch"] * VALU_CYCLES_PER_INST / (N_CU * dom_ms * 1e-3 * CLOCK_HZ))
            if issue_pmc.get("lds_active_quad_cycles_per_launch") else None,
            "wait_frac_of_wave_cycles": (issue_pmc["wait_any_quad_cycles_per_launch"] / issue_pmc["wave_quad_cycles_per_launch"])
            if issue_pmc.get("wait_any_quad_cycles_per_launch") and issue_pmc.get("wave_quad_cycles_per_launch") else None,
            "build_matches": issue_pmc.get("build_matches"),
            "profiled_build": issue_pmc.get("build"),
            "source": "profiles/%s (tag %s): PMC pass collected with tools/pmc_ab.sh, not in this run; build_matches = the digest of the "
                      "profiled path kernel equals this run's" % ("pmc_issue.json" if issue_pmc.get("workload", "c2") == "c2"
                                                                    else "pmc_issue_%s.json" % issue_pmc.get("workload"), issue_pmc.get("tag"))}


def binding_of(issue):
    issue_frac = issue["valu_util"] if issue else None
    lane_frac = issue["lanes_active"] / 64.0 if issue else None
    return {"bound": ("lds-pipe" if issue and issue.get("lds_util") and issue_frac is not None and issue["lds_util"] > issue_frac else "valu-issue"),
            "issue_frac": issue_frac, "lane_frac": lane_frac,
            "useful_lane_frac": (issue_frac * lane_frac) if issue else None,
            "lds_frac": issue["lds_util"] if issue else None,
            "wait_frac_of_wave_cycles": issue.get("wait_frac_of_wave_cycles") if issue else None}


N_SIMD, N_CU, CLOCK_HZ, VALU_CYCLES_PER_INST = 1024, 256, 2.4e9, 4      # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock; SQ counters tick in quad-cycles


def single_tick_latency(hip, lat, scen, vel, batch, n_ticks):
    """p50 / p99 of one synchronous single-scenario ltpl_tick_batch call. Inside the timer: packing the scenario into the
    ABI structs (PathsBatch / TickVelBatch), the C call (H2D + kernel + D2H) and unpacking into the reference's dicts."""
    lat_us = []
    one_res, one_vres = hip.new_paths_result(1), _capi.TickVelResult(1, hip.caps.max_path_pts)
    for i in range(100 + n_ticks):
        k = i % 64
        t1 = time.perf_counter()
        b1 = _capi.PathsBatch([scen[k]], w_last_edges=W_LAST)
        v1 = _capi.TickVelBatch(vel.params, 1, vel.vel_plan[k:k + 1], vel.vel_est[k:k + 1],
                                np.array([[vel.pos_x[k], vel.pos_y[k]]]),
                                vel.veh_vel[batch.veh_off[k]:batch.veh_off[k + 1]])
        hip.tick_batch(b1, v1, one_res, one_vres)
        one_res.action_sets(0, scen[k]['start_node'][0], lat.num_layers)
        if i >= 100:
            lat_us.append((time.perf_counter() - t1) * 1e6)
    return np.array(lat_us), (b1, v1)


def persistent_tick_latency(lat, scen, vel, batch, n_ticks, device, launched_hip):
    """The same single ticks through the RESIDENT kernel (ltpl_create_ex(LTPL_CREATE_PERSISTENT_TICK), k_tick_persistent: one workgroup
    that stays on the device and receives every tick through a mailbox in page-locked memory -- no launch, no H2D copy call, warm
    instruction cache). Own handle, closed before anything else touches the device (a resident kernel keeps device-wide synchronisations
    waiting until its idle limit). `bit_identical`: the first 64 ticks against the launched kernel's results, every output array."""
    hp = _capi.HipBackend(lat, device=device, persistent_tick=True)
    try:
        st0 = hp.persistent_stats()
        if not st0["enabled"]:
            return {"enabled": False, "why": "the lattice's single-tick kernel has no compile-time LDS plan"}
        res, vres = hp.new_paths_result(1), _capi.TickVelResult(1, hp.caps.max_path_pts)
        ref, vref = launched_hip.new_paths_result(1), _capi.TickVelResult(1, launched_hip.caps.max_path_pts)
        same = True
        lat_us = []
        for i in range(100 + n_ticks):
            k = i % 64
            t1 = time.perf_counter()
            b1 = _capi.PathsBatch([scen[k]], w_last_edges=W_LAST)
            v1 = _capi.TickVelBatch(vel.params, 1, vel.vel_plan[k:k + 1], vel.vel_est[k:k + 1], np.array([[vel.pos_x[k], vel.pos_y[k]]]),
                                    vel.veh_vel[batch.veh_off[k]:batch.veh_off[k + 1]])
            hp.tick_batch(b1, v1, res, vres)
            res.action_sets(0, scen[k]['start_node'][0], lat.num_layers)
            if i >= 100:
                lat_us.append((time.perf_counter() - t1) * 1e6)
            elif i < 64:
                launched_hip.tick_batch(b1, v1, ref, vref)
                na = int(ref.n_actions[0])
                same = same and int(res.n_actions[0]) == na and all(
                    np.array_equal(getattr(res, f)[0, :na], getattr(ref, f)[0, :na]) for f in ("action_id", "valid", "reduced", "n_nodes", "n_pts", "n_ties"))
                for a in range(na):
                    if ref.valid[0, a]:
                        n, nn = int(ref.n_pts[0, a]), int(ref.n_nodes[0, a])
                        same = same and np.array_equal(res.nodes[0, a, :nn], ref.nodes[0, a, :nn]) and np.array_equal(res.coeff[0, a, :nn - 1], ref.coeff[0, a, :nn - 1]) \
                            and np.array_equal(res.path_param[0, a, :n], ref.path_param[0, a, :n]) and np.array_equal(vres.vx[0, a, :n], vref.vx[0, a, :n]) \
                            and np.array_equal(vres.ax[0, a, :n], vref.ax[0, a, :n])
        st = hp.persistent_stats()
        lat_us = np.array(lat_us)
        return {"enabled": True, "p50": float(np.percentile(lat_us, 50)), "p99": float(np.percentile(lat_us, 99)), "mean": float(lat_us.mean()),
                "ticks": int(lat_us.size), "device_us": st["device_us_mean"], "kernel_starts": int(st["launches"]), "bit_identical": bool(same),
                "idle_limit_ms": st["idle_ms"],
                "what": "the single tick of `p50/p99` served by the resident kernel; device_us = sequence number seen .. outputs fenced out "
                        "(wall_clock64 inside the kernel, mean over all ticks)"}
    finally:
        hp.close()


def dropin_latency(hip, lat, max_ticks):
    """One closed-loop tick of the planner entry points (the C++ OnlineTrajectoryHandler behind the ABI): replay of the
    recorded C2 loop (tests/golden/c2_ticks.npz: inputs of the unmodified reference, tick by tick). Inside the timer: packing
    the objects, ltpl_planner_calc_paths, copy-out of the path dict, ltpl_planner_calc_vel_profile, copy-out of the
    trajectory set -- everything Graph_LTPL.calc_paths + calc_vel_profile hand back to the caller."""
    from oracle.fixture_io import load_records          # fixture reader only (the recording is the input stream)
    from graphbasedlocaltrajectoryplanner_amd import tick_replay as pr
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    ticks = load_records(os.path.join(ROOT, "tests", "golden", "c2_ticks.npz"))[:max_ticks]
    pl = Planner(hip, 1)
    st = ticks[0]['start']
    pl.set_start(0, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    us, keys_ok = [], True
    for t in ticks:
        veh = pr.vehicles_of_tick(t)
        zg = pr.zone_gids_of_tick(lat, t)
        va = t['vel_args']
        t1 = time.perf_counter()
        pl.calc_paths([t['action_id_sel']], [t['t']], [veh], [zg])
        pd = pl.paths(0)
        pl.calc_vel_profile([t['pos_est']], va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'],
                            local_gg=tuple(va['local_gg']), ax_max_machines=va['ax_max_machines'],
                            safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])
        traj, ids, _ = pl.trajectories(0)
        us.append((time.perf_counter() - t1) * 1e6)
        keys_ok = keys_ok and pd['keys'] == t['paths']['keys'] and list(traj.keys()) == t['vel']['keys']
    pl.close()
    us = np.array(us[100:]) if len(us) > 200 else np.array(us)
    return us, keys_ok


def closed_loop_rate(hip, lat, n_planners, n_ticks):
    """State-carrying closed loop of a BATCH of planners (ltpl_planner_*: the OnlineTrajectoryHandler state machine in C++ behind the ABI,
    one seam-(1) launch and one seam-(2) launch per tick for all planners): every planner is fed the recorded inputs of the C2 loop and
    carries its own iterative memory from tick to tick. Host-inclusive (Python packing, H2D, kernels, D2H of the path slabs, host state
    machine). Returns planner-ticks per second and whether planner 0 still offers the recorded action sets."""
    from oracle.fixture_io import load_records          # fixture reader only (the recording is the input stream)
    from graphbasedlocaltrajectoryplanner_amd import tick_replay as pr
    from graphbasedlocaltrajectoryplanner_amd.planner import Planner
    ticks = load_records(os.path.join(ROOT, "tests", "golden", "c2_ticks.npz"))[:n_ticks + 10]
    n = n_planners
    pl = Planner(hip, n)
    st = ticks[0]['start']
    for s_ in range(n):
        pl.set_start(s_, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    ok, t_sum = True, 0.0
    for k, t in enumerate(ticks):
        veh = [pr.vehicles_of_tick(t)] * n
        zg = [pr.zone_gids_of_tick(lat, t)] * n
        va = t['vel_args']
        t0 = time.perf_counter()
        pl.calc_paths([t['action_id_sel']] * n, [t['t']] * n, veh, zg)
        pl.calc_vel_profile([t['pos_est']] * n, va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                            ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])
        if k >= 10:
            t_sum += time.perf_counter() - t0
        ok = ok and list(pl.trajectories(0)[0].keys()) == t['vel']['keys']
    pl.close()
    return n * (len(ticks) - 10) / t_sum, ok


def closed_loop_device_rate(hip, lat, n_planners, n_ticks, live=True):
    """State-carrying closed loop of a FLEET (ltpl_fleet_*, ABI v5): the planners' iterative memory lives in device memory and every stage
    of the tick that ltpl_planner_* runs on the host runs as a kernel (one wave64 per planner) between the path kernel and the velocity
    kernel; the recorded inputs of the C2 loop are uploaded as a tape and all planners advance through ``n_ticks`` consecutive ticks back to
    back without host synchronisation. Returns planner-ticks per second (device time from the first to the last launch, hipEvents) and
    whether the first and the last planner reproduce the reference's recorded tick (cut indices, trajectory keys / ids, digests 1e-5)."""
    from oracle.fixture_io import load_records          # fixture reader only (the recording is the input stream and the expected output)
    from graphbasedlocaltrajectoryplanner_amd import tick_replay as pr
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    ticks = load_records(os.path.join(ROOT, "tests", "golden", "c2_ticks.npz"))[:n_ticks]
    fleet = Fleet(hip, n_planners)
    st = ticks[0]['start']
    fleet.set_start_range(0, n_planners, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    for t in ticks:
        va = t['vel_args']
        fleet.tape_append_groups([(n_planners, dict(
            prev_action=t['action_id_sel'], t_now=t['t'], vehicles=pr.vehicles_of_tick(t), zone_gids=pr.zone_gids_of_tick(lat, t),
            pos_est=t['pos_est'], vel_est=va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
            safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj']))], ax_max_machines=va['ax_max_machines'])
    ms = fleet.tape_run(0, len(ticks))
    tape_ok = True
    try:
        for p in (0, n_planners - 1):
            traj, ids, ref = fleet.trajectories(p)
            pr.check_trajectories(traj, ids, ref, ticks[-1], "planner %d" % p)
    except AssertionError as e:
        sys.stderr.write("closed_loop_device: %s\n" % e)
        tape_ok = False
    fleet.close()
    if not live:
        return n_planners * len(ticks) / (ms * 1e-3), ms / len(ticks), tape_ok, None
    # the same ticks with LIVE inputs: the host hands the fleet's inputs over every tick (arrays packed outside the timed region, as a
    # simulator that already holds its vehicles' states as arrays would), two synchronous calls per tick -- host-inclusive wall time
    fleet = Fleet(hip, n_planners)
    fleet.set_start_range(0, n_planners, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    n_live = min(len(ticks), 60)
    packed = []
    for t in ticks[:n_live]:
        va = t['vel_args']
        packed.append(fleet.pack_groups([(n_planners, dict(
            prev_action=t['action_id_sel'], t_now=t['t'], vehicles=pr.vehicles_of_tick(t), zone_gids=pr.zone_gids_of_tick(lat, t),
            pos_est=t['pos_est'], vel_est=va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
            safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj']))], ax_max_machines=va['ax_max_machines']))
    t_live, live_ok = 0.0, True
    for k, (pi, vi, _keep) in enumerate(packed):
        t0 = time.perf_counter()
        fleet.calc_paths_packed(pi)
        fleet.calc_vel_profile_packed(vi)
        if k >= 5:
            t_live += time.perf_counter() - t0
    try:
        traj, ids, ref = fleet.trajectories(n_planners - 1)
        pr.check_trajectories(traj, ids, ref, ticks[n_live - 1], "live planner %d" % (n_planners - 1))
    except AssertionError as e:
        sys.stderr.write("closed_loop_device (live inputs): %s\n" % e)
        live_ok = False
    fleet.close()
    live_rate = n_planners * max(n_live - 5, 1) / max(t_live, 1e-9)
    return n_planners * len(ticks) / (ms * 1e-3), ms / len(ticks), tape_ok and live_ok, live_rate


def fleet_group_inputs(pr, lat, t):
    va = t['vel_args']
    return dict(prev_action=t['action_id_sel'], t_now=t['t'], vehicles=pr.vehicles_of_tick(t), zone_gids=pr.zone_gids_of_tick(lat, t),
                pos_est=t['pos_est'], vel_est=va['vel_est'], vel_max=va['vel_max'], gg_scale=va['gg_scale'], local_gg=tuple(va['local_gg']),
                safety_d=va['safety_d'], incl_emerg_traj=va['incl_emerg_traj'])


def closed_loop_device_mixed(hip, lat, n_planners, n_ticks, names=("c2", "overtake", "zonewall", "c1")):
    """The fleet on a MIXED tape: the planners are split into one group per recording of the reference (C2 std loop, overtakes with
    dropped keys and emergency profiles, zone wall with reduced horizons, C1 static obstacle + wall), so that neighbouring waves run
    different branches of the state machine on different data. Checked: the first, the middle and the last planner of EVERY group at
    three ticks of the run against what the unmodified reference produced on that tick (cut indices, keys / ids, digests 1e-5, full
    trajectories where the recording holds them)."""
    from oracle.fixture_io import load_records          # fixture reader only (input streams and expected outputs)
    from graphbasedlocaltrajectoryplanner_amd import tick_replay as pr
    from graphbasedlocaltrajectoryplanner_amd.fleet import Fleet
    recs = [load_records(os.path.join(ROOT, "tests", "golden", "%s_ticks.npz" % nm))[:n_ticks] for nm in names]
    n_ticks = min(len(r) for r in recs)
    sizes = [n_planners // len(names)] * len(names)
    sizes[0] += n_planners - sum(sizes)
    fleet = Fleet(hip, n_planners)
    for k in range(n_ticks):
        fleet.tape_append_groups([(sz, fleet_group_inputs(pr, lat, ticks[k])) for sz, ticks in zip(sizes, recs)],
                                 ax_max_machines=recs[0][k]['vel_args']['ax_max_machines'])
    p = 0
    for sz, ticks in zip(sizes, recs):
        st = ticks[0]['start']
        fleet.set_start_range(p, p + sz, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
        p += sz
    from graphbasedlocaltrajectoryplanner_amd.planner import KEY_IDS
    stops = sorted(set([max(1, n_ticks // 3), max(1, 2 * n_ticks // 3), n_ticks]))
    ms, ok, checked, t_prev, digested = 0.0, True, 0, 0, 0
    for t_stop in stops:
        ms += fleet.tape_run(t_prev, t_stop - t_prev)
        t_prev = t_stop
        # EVERY planner of the fleet against the recording of its group, digested on the device (ltpl_fleet_digest: cut indices, velocity
        # plan, keys, ids, rows, s_end, vx[0], vx[-1], sum of vx per trajectory) ...
        if hasattr(fleet, "digest") and hasattr(getattr(fleet, "lib", None), "ltpl_fleet_digest"):
            dig = fleet.digest()
            p = 0
            for sz, ticks, nm in zip(sizes, recs, names):
                try:
                    digested += pr.check_digests(dig[p:p + sz], ticks[t_stop - 1], KEY_IDS, "%s tick %d (first planner of the group: %d)" % (nm, t_stop - 1, p))
                except AssertionError as e:
                    sys.stderr.write("closed_loop_device_mixed: %s\n" % e)
                    ok = False
                p += sz
        # ... and full trajectories (where the recording holds them) of the first, the middle and the last planner of every group
        p = 0
        for sz, ticks, nm in zip(sizes, recs, names):
            for q in sorted(set((p, p + sz // 2, p + sz - 1))):
                try:
                    traj, ids, ref = fleet.trajectories(q)
                    pr.check_trajectories(traj, ids, ref, ticks[t_stop - 1], "%s tick %d planner %d" % (nm, t_stop - 1, q))
                    checked += 1
                except AssertionError as e:
                    sys.stderr.write("closed_loop_device_mixed: %s\n" % e)
                    ok = False
            p += sz
    fleet.close()
    return {"planner_ticks_per_s": n_planners * n_ticks / (ms * 1e-3), "planners": n_planners, "ticks": n_ticks,
            "ms_per_fleet_tick": ms / n_ticks, "groups": list(names), "matches_recording": ok, "planner_checks": checked,
            "planners_digested_on_device": digested, "checked_at_ticks": [t - 1 for t in stops],
            "what": "ltpl_fleet_* on a mixed tape: %d groups of planners replay different recordings of the reference side by side (device "
                    "time of the tape segments, HIP events); at three ticks EVERY planner's digest (computed on the device) and the full "
                    "trajectories of the first / middle / last planner of every group are compared with the reference's recording" % len(names)}


def sub_batch(scen, batch, vel, lo, hi):
    """Scenarios [lo, hi) of a generated batch as their own ABI structs."""
    b = _capi.PathsBatch(scen[lo:hi], w_last_edges=W_LAST)
    vo = np.asarray(batch.veh_off)
    v = _capi.TickVelBatch(vel.params, hi - lo, vel.vel_plan[lo:hi], vel.vel_est[lo:hi],
                           np.column_stack((vel.pos_x[lo:hi], vel.pos_y[lo:hi])), vel.veh_vel[int(vo[lo]):int(vo[hi])])
    return b, v


def c4_legs(hip, scen, batch, vel, total=1024, n_gpus=8, min_s=0.5):
    """BASELINE config C4 AS STATED ("batch of 1024 independent obstacle scenarios on the Monteblanco lattice, sharded 8 x MI355X"; SURVEY
    section 8d C4: 128 per GPU) on ONE GPU: the whole batch in one call, and the 128-scenario shard one of 8 GPUs would run -- resident
    (inputs in HBM, ltpl_batch_run) and PCIe-inclusive (ltpl_tick_batch_compact: host buffers in, packed trajectories out). 128 waves on
    256 CUs is a latency run, not a throughput run: us per call is the figure."""
    out = {"what": "C4 as BASELINE states it: %d scenarios in total; `one_gpu` = all of them in one call, `shard` = the %d scenarios one of %d "
                   "GPUs owns (the 8-GPU job's time per step is the shard's: no collective on the data path). resident_* = inputs in HBM, "
                   "back-to-back steps; pcie_* = ltpl_tick_batch_compact per call, host wall time. The library serves batches of up to two fused "
                   "workgroups per compute unit (512 scenarios on the MI355X) with the fused tick kernel -- one launch -- and larger ones with the "
                   "one-wave pipeline (round 6, profiles/r06h_c4_fused_ab.txt): the shard runs fused, the whole batch on one GPU the pipeline"
                   % (total, total // n_gpus, n_gpus)}
    total = min(total, len(scen))
    for key, n in (("one_gpu", total), ("shard", max(1, total // n_gpus))):
        b, v = sub_batch(scen, batch, vel, 0, n)
        hip.batch_upload(b, v)
        hip.batch_run(reps=20, timed=False)
        reps = 100
        t0 = time.perf_counter(); hip.batch_run(reps=reps, timed=True); el = time.perf_counter() - t0
        reps = max(reps, int(math.ceil(min_s / max(el / reps, 1e-7))))
        t0 = time.perf_counter(); hip.batch_run(reps=reps, timed=True); el = time.perf_counter() - t0
        comp = hip.new_compact_trajectories(n, max_rows=115)
        hip.tick_batch_compact(b, v, comp)
        us = []
        for _ in range(30):
            t1 = time.perf_counter()
            hip.tick_batch_compact(b, v, comp)
            us.append((time.perf_counter() - t1) * 1e6)
        us = np.array(us)
        out[key] = {"scenarios": n, "resident_ticks_per_s": n * reps / el, "resident_us_per_step": el / reps * 1e6, "resident_steps": reps,
                    "pcie_ticks_per_s": n / (float(np.median(us)) * 1e-6), "pcie_us_per_call_p50": float(np.median(us)),
                    "pcie_us_per_call_p99": float(np.percentile(us, 99))}
    out["projected_8gpu_ticks_per_s"] = {"resident": n_gpus * out["shard"]["resident_ticks_per_s"], "pcie": n_gpus * out["shard"]["pcie_ticks_per_s"],
                                         "basis": "8 x the shard's rate on this GPU (independent shards, no exchange): a projection, NOT a measurement"}
    return out


def c5_latency(horizon_m, n_ticks, device=0):
    """BASELINE config C5: high-resolution oval (0.5 m layer spacing, 21 lateral nodes), a slow opponent ahead so that the follow-mode
    velocity profile runs on every tick; single-scenario synchronous ltpl_tick_batch calls, host wall time including marshalling and
    PCIe. horizon 300 m = 600 layers (long-horizon mode), 100 m = the fused LDS-resident kernel."""
    from graphbasedlocaltrajectoryplanner_amd.scenario_gen import raceline_state
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c5_lattice
    lat = c5_lattice(horizon=float(horizon_m))
    hip = _capi.HipBackend(lat, device=device)
    rng = np.random.default_rng(2)
    singles = []
    params = _capi.VelParamSet(len_veh=lat.veh_length)
    for _ in range(64):
        sl = int(rng.integers(0, lat.num_layers)); sn = int(lat.raceline_index[sl])
        x, y, psi, v = raceline_state(lat, float(lat.s_raceline[sl]) + rng.uniform(20.0, 80.0))
        v = float(v) * rng.uniform(0.2, 0.5)
        pred = np.array([[x - np.sin(psi) * v * 0.2, y + np.cos(psi) * v * 0.2]])
        sc = {"start_node": (sl, sn), "action_sets": True, "vehicles": [(2.5, np.vstack((np.array([[x, y]]), pred)))], "zone_gids": [],
              "last_nodes": None, "obj_in_const": False, "obj_besides": False, "last_action": None, "const_closest": None,
              "psi_s": float(lat.node_psi[lat.layer_off[sl] + sn])}
        b1 = _capi.PathsBatch([sc], w_last_edges=W_LAST)
        vp = float(rng.uniform(5.0, 35.0))
        v1 = _capi.TickVelBatch(params, 1, [vp], [vp], lat.node_pos[lat.layer_off[sl] + sn][None, :], np.array([v]))
        singles.append((b1, v1))
    res, vres = hip.new_paths_result(1), _capi.TickVelResult(1, hip.caps.max_path_pts)
    lat_us, n_follow = [], 0
    for i in range(50 + n_ticks):
        b1, v1 = singles[i % 64]
        t1 = time.perf_counter()
        hip.tick_batch(b1, v1, res, vres)
        if i >= 50:
            lat_us.append((time.perf_counter() - t1) * 1e6)
            n_follow += int(((res.action_id == _capi.ACT_FOLLOW) & (res.valid == 1)).sum())
    lat_us = np.array(lat_us)
    out = {"config": "C5: high-res oval, %d layers x %d nodes, %d edges, horizon %d layers, %d path samples, follow profile on %.0f %% of "
                     "the ticks" % (lat.num_layers, int(lat.nodes_in_layer.max()), lat.num_edges, hip.caps.max_path_nodes,
                                    hip.caps.max_path_pts, 100.0 * n_follow / lat_us.size),
           "p50_us": float(np.percentile(lat_us, 50)), "p99_us": float(np.percentile(lat_us, 99)), "mean_us": float(lat_us.mean()),
           "ticks": int(lat_us.size), "us_per_layer_p50": float(np.percentile(lat_us, 50)) / max(hip.caps.max_path_nodes, 1)}
    hip.close()
    return out


def c3_throughput(n_batch, device=0, min_s=1.5, n_parity=256):
    """BASELINE config C3 (the "HBM roofline run"): synthetic oval, 400 layers x 25 nodes, ~98 k edges, 32 static obstacles (64 obstacle
    positions) per scenario; resident tick pipeline like the headline, its own roofline figures and a parity check against the oracle."""
    from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice
    lat = c3_lattice()
    hip = _capi.HipBackend(lat, device=device)
    scen, batch, vel = make_batch(lat, n_batch, seed=1, workload="c3")
    hip.batch_upload(batch, vel)
    hip.batch_run(reps=5, timed=False)
    reps = 20
    t0 = time.perf_counter(); hip.batch_run(reps=reps, timed=True); el = time.perf_counter() - t0
    reps = max(reps, int(math.ceil(min_s / max(el / reps, 1e-6))))
    t0 = time.perf_counter(); hip.batch_run(reps=reps, timed=True); el = time.perf_counter() - t0
    paths_ms = hip.batch_last_paths_ms()
    res, vres = hip.batch_download()
    prof_ms = hip.batch_run_profile(reps=10)
    ab = algorithmic_bytes(lat, batch, res)
    ab_paths = ab["mask"] + ab["sweep"] + ab["path"]
    dom_ms = paths_ms if paths_ms > 0.0 else prof_ms[0]
    out = {"ticks_per_s": n_batch * reps / el, "batch": n_batch, "steps": reps, "ms_per_step": el / reps * 1e3,
           "kernel_ms": dom_ms, "kernel_ms_not_overlapped": prof_ms[0],
           "roofline_frac": ab_paths / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "roofline_frac_refline_per_position": (ab_paths - ab["mask"] + ab["mask_refline_per_position"]) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "algorithmic_bytes_per_tick": ab["total"] / n_batch, "split_per_tick": {k: ab[k] / n_batch for k in ("mask", "sweep", "path", "vel")},
           "workload": "C3: synthetic oval lattice (%d layers / %d nodes / %d edges), 4 action primitives, 32 static obstacles (64 obstacle "
                       "positions); %d scenarios per step" % (lat.num_layers, lat.num_nodes, lat.num_edges, n_batch)}
    # what binds C3 (BASELINE's "HBM roofline run"): the same counter passes as the headline, collected on the C3 batch
    # (PMC_WORKLOAD=c3 tools/pmc_ab.sh -> profiles/pmc_issue_c3.json, tools/profile_summarise.py <tag> ... c3 -> profiles/pmc_traffic_c3.json)
    stamp = library_stamp(hip)
    issue = issue_summary(read_issue(n_batch, "c3", stamp), n_batch, dom_ms)
    tr = read_traffic(n_batch, "c3", stamp)
    out["library_kernel"] = {k: stamp.get(k) for k in ("kernel_symbol", "isa_sha256", "resources")}
    out["binding"] = binding_of(issue) if issue else None
    out["issue"] = issue
    out["traffic"] = tr.get("hbm_bytes_per_launch") if tr else None
    out["traffic_frac"] = (tr["hbm_bytes_per_launch"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if tr and tr.get("hbm_bytes_per_launch") else None
    out["traffic_build_matches"] = tr.get("build_matches") if tr else None
    idx = sample_indices(n_batch, n_parity)
    _, ref = cpu_baseline(lat, scen, batch, vel, idx) if n_parity > 0 else (None, None)
    if ref is not None:
        ok, detail = parity_check(res, vres, ref, idx)
        out["parity_checked"] = bool(ok)
        out["parity_detail"] = {k: detail[k] for k in ("scenarios", "paths", "max_rel_err", "integer_mismatches", "elementwise_rel_err")}
    hip.close()
    return out


def worker(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # one process per GPU. LTPL_BENCH_SHARE_GPU=1 (test only: rehearses the N > 1 code path on a single-GPU box) lets all
    # ranks use device 0 and swaps RCCL for gloo, which accepts ranks that share a device.
    share = os.environ.get("LTPL_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    if not share and dev_index >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible" % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        world = dist.get_world_size()                      # the rank count the collective backend actually sees

    if args.workload == "c3":
        from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice
        lat = c3_lattice()
    else:
        lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    hip = _capi.HipBackend(lat, device=dev_index)
    strong = getattr(args, "scaling", "weak") == "strong"
    if strong:
        # STRONG scaling (BASELINE config C4: a FIXED batch of --batch-total scenarios, block-partitioned over the ranks): every rank generates
        # the same batch (same seed) and keeps its block [lo, hi) (sharding.shard_bounds) -- no collective on the data path, results stay on
        # the ranks
        from graphbasedlocaltrajectoryplanner_amd.sharding import shard_bounds
        scen_all, batch_all, vel_all = make_batch(lat, args.batch_total, seed=1, workload=args.workload)
        lo, hi = shard_bounds(args.batch_total, rank, world)
        if hi <= lo:
            raise SystemExit("bench.py: --batch-total %d leaves rank %d of %d without a scenario" % (args.batch_total, rank, world))
        scen = scen_all[lo:hi]
        batch, vel = sub_batch(scen_all, batch_all, vel_all, lo, hi)
        args.batch = hi - lo
    else:
        scen, batch, vel = make_batch(lat, args.batch, seed=1 + rank, workload=args.workload)
    hip.batch_upload(batch, vel)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # warm-up (untimed), then one untimed K-step block to size the timed region (>= MIN_TIMED_S; agreed over the ranks)
    for _ in range(args.warmup):
        hip.batch_run(reps=1, timed=False)
    barrier()
    t0 = time.perf_counter()
    hip.batch_run(reps=args.steps, timed=False)
    barrier()
    block_s = max_over_ranks(time.perf_counter() - t0)
    inner = max(1, int(math.ceil(MIN_TIMED_S / max(block_s, 1e-6)))) if not args.exact_steps else 1
    timed_steps = args.steps * inner
    barrier()
    t0 = time.perf_counter()
    ms_kernel = hip.batch_run(reps=timed_steps, timed=True)     # HIP events on the library's stream + wait
    own_s = time.perf_counter() - t0                             # this rank's own time for its shard (before the closing barrier)
    paths_ms_live = hip.batch_last_paths_ms()                    # path kernel inside the timed region (events per launch)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    res, vres = hip.batch_download()

    def gather_over_ranks(x):
        if dist is None:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if share else "cuda")
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        return [float(o.item()) for o in outl]
    per_rank_ms = [v / timed_steps * 1e3 for v in gather_over_ranks(own_s)]
    # N > 1: the state-carrying fleet sharded like the scenario batches -- every rank owns a block of the vehicles (sharding.fleet_shard),
    # no communication between the fleets; a short tape of the recorded C2 loop per rank, rates summed over the ranks
    fleet_sharded = None
    if world > 1 and args.workload == "c2" and not args.no_extra:
        from graphbasedlocaltrajectoryplanner_amd.sharding import fleet_shard
        lo, hi = fleet_shard(getattr(args, 'fleet_planners', 32768), rank, world)
        rate, ms_tick, ok_f, _ = closed_loop_device_rate(hip, lat, hi - lo, min(getattr(args, 'fleet_ticks', 200), 60), live=False)
        rates = gather_over_ranks(rate)
        oks = gather_over_ranks(1.0 if ok_f else 0.0)
        fleet_sharded = {"planner_ticks_per_s": float(sum(rates)), "per_rank_planner_ticks_per_s": rates,
                         "planners_total": getattr(args, 'fleet_planners', 32768), "matches_recording": bool(min(oks) > 0.5),
                         "what": "one fleet per rank on its block of the vehicles (sharding.fleet_shard), no collective on the data path"}
        hip.batch_upload(batch, vel)

    # per-kernel durations of the pipeline (HIP events between the launches on the library's stream), outside the timed region
    prof_ms = hip.batch_run_profile(reps=20)
    # units of one step over ALL ranks (weak: every rank owns --batch scenarios; strong: the fixed --batch-total)
    total_units = args.batch_total if strong else world * args.batch
    if rank == 0:
        ab = algorithmic_bytes(lat, batch, res)
        kern_ms = ms_kernel / timed_steps
        n_paths = int(res.valid.sum())
        # dominant kernel = the path kernel (mask + sweeps + spline); its algorithmic bytes exclude the velocity stage
        ab_paths = ab["mask"] + ab["sweep"] + ab["path"]
        dom_ms = paths_ms_live if paths_ms_live > 0.0 else prof_ms[0]
        achieved = ab_paths / (dom_ms * 1e-3) / 1e9
        extra = {}
        lat_us, device_us, drop_us, drop_ok, persistent = np.zeros(0), None, np.zeros(0), None, None
        # N > 1 is the scaling run: whole-job throughput only. The single-GPU legs (latency, extras, the CPU baseline and the parity
        # full-size CPU baseline) belong to the N = 1 line -- the other ranks would sit in the final barrier while rank 0 runs them
        solo = world == 1
        if args.latency_ticks > 0 and solo:
            lat_us, single = single_tick_latency(hip, lat, scen, vel, batch, args.latency_ticks)
            # device-only time of one single-scenario tick (latency method): the same fused kernel launched
            # back to back on a device-resident scenario, HIP events on the library's stream
            hip.batch_upload(single[0], single[1])
            hip.batch_run(reps=20, timed=False)
            device_us = hip.batch_run(reps=200, timed=True) / 200 * 1e3
            persistent = persistent_tick_latency(lat, scen, vel, batch, args.latency_ticks, dev_index, hip)
            if args.workload == "c2":
                drop_us, drop_ok = dropin_latency(hip, lat, args.dropin_ticks)
        if args.workload == "c2" and not args.no_extra and solo:
            # companion number: every scenario has an opponent 20-80 m ahead, so the [follow, left, right] template is live
            n3 = min(args.batch, 8192)
            scen3, batch3, vel3 = make_batch(lat, n3, seed=1001, workload="c2_near")
            hip.batch_upload(batch3, vel3)
            hip.batch_run(reps=5, timed=False)
            t3 = time.perf_counter()
            reps3 = 60
            hip.batch_run(reps=reps3, timed=True)
            el3 = time.perf_counter() - t3
            res3, _ = hip.batch_download()
            extra["three_slot_ticks_per_s"] = n3 * reps3 / el3
            extra["three_slot_paths_per_tick"] = float(res3.valid.sum()) / n3
            extra["three_slot_workload"] = ("C2 with the nearest opponent 20-80 m ahead of the ego in every scenario, %d scenarios "
                                            "per step x %d steps" % (n3, reps3))
            # PCIe-inclusive rate of the host-buffer entry point (packing + H2D + kernels + D2H + scatter per call)
            npc = min(args.batch, 8192)
            bpc = _capi.PathsBatch(scen[:npc], w_last_edges=W_LAST)
            nv = int(bpc.veh_off[-1])
            vpc = _capi.TickVelBatch(vel.params, npc, vel.vel_plan[:npc], vel.vel_est[:npc],
                                     np.column_stack((vel.pos_x[:npc], vel.pos_y[:npc])), vel.veh_vel[:nv])
            rpc, vrpc = hip.new_paths_result(npc), _capi.TickVelResult(npc, hip.caps.max_path_pts)
            hip.tick_batch(bpc, vpc, rpc, vrpc)
            tp = time.perf_counter()
            for _ in range(5):
                hip.tick_batch(bpc, vpc, rpc, vrpc)
            slab_rate = npc * 5 / (time.perf_counter() - tp)
            comp = hip.new_compact_trajectories(npc, max_rows=115)
            hip.tick_batch_compact(bpc, vpc, comp)
            tp = time.perf_counter()
            for _ in range(10):
                hip.tick_batch_compact(bpc, vpc, comp)
            extra["pcie_inclusive"] = {"ticks_per_s": npc * 10 / (time.perf_counter() - tp), "scenarios_per_call": npc,
                                       "rows_per_trajectory": 115, "bytes_out_per_tick": float(comp.struct.total_rows) * 56 / npc,
                                       "capacity_slab_ticks_per_s": slab_rate,
                                       "what": "ltpl_tick_batch_compact: host buffers in, trajectories [s,x,y,psi,kappa,vx,ax] "
                                               "(115 export rows, Graph_LTPL.py:401-406) packed on the device and DMA-written into "
                                               "page-locked host memory; capacity_slab_* = ltpl_tick_batch with full capacity slabs"}
            # state-carrying closed loop of a batch of planners (host state machine + two launches per tick): host-bound, reported as is
            clr, clok = closed_loop_rate(hip, lat, 256, 60)
            extra["closed_loop"] = {"planner_ticks_per_s": clr, "planners": 256, "ticks": 60, "keys_match_recording": clok,
                                    "what": "256 planners x 60 consecutive ticks through ltpl_planner_calc_paths / _calc_vel_profile, every "
                                            "planner carrying its own iterative memory; host-inclusive (packing, PCIe, host state machine)"}
            # the same closed loop with the planners' state in device memory (fleet): no host work per planner
            n_fp, n_ft = getattr(args, 'fleet_planners', 32768), getattr(args, 'fleet_ticks', 200)
            cdr, cd_ms, cdok, cd_live = closed_loop_device_rate(hip, lat, n_fp, n_ft)
            extra["closed_loop_device_mixed"] = closed_loop_device_mixed(hip, lat, n_fp, n_ft)
            if n_fp > 8192:            # the fleet size the rounds before quoted (wave-per-job follow kernel below 12 288 planners)
                m8 = closed_loop_device_mixed(hip, lat, 8192, n_ft)
                extra["closed_loop_device_mixed"]["at_8192_planners"] = {k: m8[k] for k in ("planner_ticks_per_s", "ms_per_fleet_tick", "matches_recording",
                                                                                             "planners_digested_on_device")}
            extra["closed_loop_device"] = {"planner_ticks_per_s": cdr, "planners": n_fp, "ticks": n_ft, "ms_per_fleet_tick": cd_ms,
                                           "inputs": "ALL planners replay the SAME recorded inputs (no divergence between the waves, identical "
                                                     "data in the caches); closed_loop_device_mixed runs four different recordings side by side",
                                           "live_inputs_planner_ticks_per_s": cd_live,
                                           "matches_recording": cdok,
                                           "what": "ltpl_fleet_*: planners with device-resident iterative memory replay consecutive ticks of the "
                                                   "reference's C2 recording from a pre-uploaded tape (state machine stages as kernels, one wave "
                                                   "per planner, around k_paths / k_vel_profile; device time, no host synchronisation inside); "
                                                   "first and last planner checked against the recording's last tick. live_inputs_*: "
                                                   "the same loop through the per-call entry points, the host hands every tick's inputs "
                                                   "over (pre-packed arrays -> page-locked staging -> H2D) and synchronises twice per tick; "
                                                   "host wall time"}
        if args.workload == "c2" and not args.no_extra and solo:
            # the other BASELINE configurations in the SAME run (own handles on the same GPU, after the headline's timed region):
            # C3 = the "HBM roofline run" (throughput, roofline fraction, parity), C5 = the latency run (both horizons)
            extra["c4"] = c4_legs(hip, scen, batch, vel, total=min(1024, args.batch))
            hip.batch_upload(batch, vel)                          # (drop the big resident sets of the legs above before the next handles)
            extra["c3"] = c3_throughput(args.c3_batch, device=dev_index)
            extra["c5"] = {"horizon_300m": c5_latency(300, args.c5_ticks, device=dev_index),
                           "horizon_100m": c5_latency(100, args.c5_ticks, device=dev_index),
                           "what": "single-scenario synchronous ltpl_tick_batch calls on the high-resolution oval (0.5 m layer spacing), host wall "
                                   "time incl. marshalling and PCIe; 300 m = 600 layers (long-horizon mode), 100 m = the fused LDS-resident kernel"}
        stamp = library_stamp(hip)
        traffic_pmc = read_traffic(args.batch, args.workload, stamp)   # measured HBM bytes per launch of the dominant kernel (PMC), or None
        traffic = traffic_pmc.get("hbm_bytes_per_launch") if traffic_pmc else None
        issue_pmc = read_issue(args.batch, args.workload, stamp)
        issue = issue_summary(issue_pmc, args.batch, dom_ms)
        # the roofline that BINDS: vector-instruction issue (useful lane-operations / peak lane-operations of the launch) next to the
        # contract's HBM figure
        binding = binding_of(issue)
        out = {
            "metric": "planning ticks/s (all action primitives), " + ("Monteblanco lattice" if args.workload == "c2" else "synthetic C3 lattice"),
            "value": total_units * timed_steps / elapsed,
            "unit": "ticks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "timed_steps": timed_steps,
            "ms_per_step": elapsed / timed_steps * 1e3,
            # every rank's own time per step for ITS shard (the line's ms_per_step is the max-over-ranks region incl. the closing barrier);
            # scaling efficiency is the driver's to compute from the per-N lines -- the field is a placeholder it may fill
            "per_rank_ms_per_step": per_rank_ms, "efficiency_vs_n1": None,
            "higher_is_better": True, "scaling": "strong" if strong else "weak",
            # BASELINE.md holds no published number for this metric (the reference publishes none); the only figure it states for this
            # config is BASELINE.json's target of >= 10 000 planning ticks/s on one GPU -- the ratio below is against that TARGET
            # BASELINE.md holds no published number for this metric (the reference publishes none): null, as the contract asks. The ratio to
            # BASELINE.json's stated TARGET (>= 10 000 ticks/s per GPU on C2) is `vs_target`; the measured CPU baseline is in `cpu_baseline`
            "vs_baseline": None,
            "vs_target": (total_units * timed_steps / elapsed) / (TARGET_TICKS_PER_S * world) if args.workload == "c2" else None,
            "vs_target_basis": "BASELINE.json target: >= 10 000 ticks/s per GPU on C2 (no published reference number exists)",
            # the path kernel (mask, sweeps, spline) computes in fp64 throughout; the lane kernels of the velocity stage read |kappa| and the
            # element length as an fp32 pair and take 1 / |kappa| from v_rcp_f32 -- their profile state and every output are fp64
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (("C2: Monteblanco lattice (%d layers / %d nodes / %d edges), 4 action primitives, 8 dynamic "
                                     "opponents (16 obstacle positions), sample zone" if args.workload == "c2" else
                                     "C3: synthetic oval lattice (%d layers / %d nodes / %d edges), 4 action primitives, 32 static "
                                     "obstacles (64 obstacle positions)") % (lat.num_layers, lat.num_nodes, lat.num_edges))
                                   + ("; a fixed batch of %d independent scenarios per step block-partitioned over the ranks (C4), tick pipeline "
                                      "(paths + velocity)" % total_units if strong else
                                      "; %d independent scenarios per GPU per step, tick pipeline (paths + velocity)" % args.batch),
                       "batch_per_gpu": args.batch, "batch_total": total_units,
                       "parallelism": "scenario-sharded x%d (no collective; results stay on the ranks)" % world},
            # the library this line was measured on (LTPL_HIP_LIB can redirect it) and the digest the counter passes are matched against
            "library": stamp,
            # `bound`: what binds the kernel (the same word as `binding.bound`: latency-bound, its busiest pipe named); `contract_bound`: the
            # roofline `achieved` / `peak` / `frac` are priced against, as the bench contract asks (no dense contraction: HBM, not MFMA)
            "roofline": {"bound": "latency / %s" % binding["bound"], "contract_bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         # what actually limits the kernel: instruction issue + dependent LDS round trips, not DRAM (the lattice is cache
                         # resident: traffic_frac ~ 0.1). `issue` = the bound that binds, from a PMC pass of this build
                         "limiter": "latency / %s (cache-resident working set)" % binding["bound"],
                         "traffic_build_matches": traffic_pmc.get("build_matches") if traffic_pmc else None,
                         "binding": dict(binding,
                                     what="issue_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz); lane_frac = "
                                          "active lanes per vector instruction / 64; useful_lane_frac = their product = share of the chip's "
                                          "lane throughput that does work; lds_frac = SQ_ACTIVE_INST_LDS x 4 / (256 CUs x kernel time x 2.4 GHz); bound = the larger of issue_frac and lds_frac. "
                                          "Both are lower bounds: the clock under this load is below the 2.4 GHz peak"),
                         "frac_refline_per_position": (ab_paths - ab["mask"] + ab["mask_refline_per_position"]) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "issue": issue,
                         "kernel": "k_paths<1>", "kernel_ms": dom_ms,
                         "kernel_ms_not_overlapped": prof_ms[0],
                         "algorithmic_bytes_per_launch": ab_paths,
                         "algorithmic_bytes_per_tick": ab["total"] / args.batch,
                         "split_per_tick": {k: ab[k] / args.batch for k in ("mask", "sweep", "path", "vel")},
                         "survey_model_per_tick": {"mask": ab["mask_survey"] / args.batch, "sweep": ab["sweep_survey"] / args.batch},
                         "mask_counts_per_tick": {"window_edges": ab["window_edges"] / args.batch, "shell_edges": ab["shell_edges"] / args.batch,
                                                  "shell_samples": ab["shell_samples"] / args.batch},
                         "pipeline_ms": {"k_paths": prof_ms[0], "k_follow_prep": prof_ms[1], "k_vel_lanes": prof_ms[2],
                                         "all_kernels_back_to_back": kern_ms},
                         "whole_tick_achieved": ab["total"] / (kern_ms * 1e-3) / 1e9,
                         "note": "achieved = algorithmic bytes of THIS algorithm / kernel time (graphbasedlocaltrajectoryplanner_amd/roofline.py: "
                                 "capsule record per window edge + samples of the shell edges the cull cannot decide, counted on the batch with "
                                 "the kernel's decision arithmetic; edge records once per scenario; round 4: the reference line once per "
                                 "scenario -- it is staged in LDS -- instead of once per obstacle position, `frac_refline_per_position` keeps the "
                                 "round-3 figure). `bound` is the contract's vocabulary (no dense contraction: HBM); the lattice is cache "
                                 "resident, measured HBM traffic is `traffic` (traffic_frac of the peak), what binds is in `binding`"},
            "latency_us": {"p50": float(np.percentile(lat_us, 50)) if lat_us.size else None,
                           "p99": float(np.percentile(lat_us, 99)) if lat_us.size else None,
                           "mean": float(lat_us.mean()) if lat_us.size else None, "ticks": int(lat_us.size),
                           "device_us": device_us,
                           "dropin_p50": float(np.percentile(drop_us, 50)) if drop_us.size else None,
                           "dropin_p99": float(np.percentile(drop_us, 99)) if drop_us.size else None,
                           "dropin_mean": float(drop_us.mean()) if drop_us.size else None,
                           "dropin_ticks": int(drop_us.size), "dropin_keys_match_recording": drop_ok,
                           "persistent_tick": persistent,
                           "what": "p50/p99: one scenario per ltpl_tick_batch call, host wall time incl. packing into the ABI structs, "
                                   "PCIe both ways and unpacking into the reference's dict structures; device_us: the tick kernel "
                                   "alone (HIP events, back-to-back launches); dropin_*: one closed-loop tick of the planner entry "
                                   "points (calc_paths + calc_vel_profile + copy-out) replaying the recorded C2 loop"},
            "paths_per_tick": n_paths / args.batch,
            "extra": extra,
        }
        if fleet_sharded is not None:
            out["extra"]["closed_loop_device_sharded"] = fleet_sharded
        if not args.no_cpu:
            ns = min(args.cpu_sample if solo else min(args.cpu_sample, 256), args.batch)      # (N > 1: a short sample, for the parity check of rank 0's shard)
            idx = sample_indices(args.batch, ns)
            out["cpu_baseline"], ref = cpu_baseline(lat, scen, batch, vel, idx)
            ok, detail = parity_check(res, vres, ref, idx)
            out["parity_checked"] = bool(ok)
            out["parity_detail"] = detail
            out["extra"]["vs_cpu_port_1core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _spawned(local_rank, args, port):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(args.gpus),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    worker(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32768, help="scenarios per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--latency-ticks", type=int, default=2000)
    ap.add_argument("--dropin-ticks", type=int, default=2500)
    ap.add_argument("--fleet-planners", type=int, default=32768, help="extra.closed_loop_device*: planners of the fleet (round 5: 32 768 like the "
                    "headline's batch -- a fleet tick is a chain of kernels with one wave per planner, 8 192 waves leave the chip half empty; the "
                    "8 192-planner rate of the earlier rounds is reported next to it)")
    ap.add_argument("--fleet-ticks", type=int, default=200, help="extra.closed_loop_device: consecutive ticks")
    ap.add_argument("--c3-batch", type=int, default=32768, help="extra.c3: scenarios per step (one size for builder and driver since round 5: the headline's)")
    ap.add_argument("--c5-ticks", type=int, default=400, help="extra.c5: ticks per horizon")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--exact-steps", action="store_true", help="time exactly --steps steps (no repetition up to 2 s)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's scaling run): --batch scenarios per GPU; strong: a fixed batch of --batch-total scenarios "
                         "block-partitioned over the ranks (BASELINE config C4: 1024 scenarios over 8 GPUs)")
    ap.add_argument("--batch-total", type=int, default=1024, help="--scaling strong: scenarios per step over all ranks")
    ap.add_argument("--workload", choices=("c2", "c3"), default="c2",
                    help="c2 = BASELINE config the metric is quoted on (default); c3 = synthetic 10k-node / 98k-edge lattice "
                         "with 32 obstacles per scenario (HBM-roofline run, reported separately under profiles/)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, world_env))
        worker(args)
    elif args.gpus == 1:
        worker(args)
    else:
        # no launcher: start one rank per GPU ourselves (fails loudly if the node has fewer GPUs)
        import socket
        import torch
        import torch.multiprocessing as mp
        share = os.environ.get("LTPL_BENCH_SHARE_GPU") == "1"
        if not share and torch.cuda.device_count() < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)


if __name__ == "__main__":
    main()
