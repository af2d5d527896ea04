"""CPU: dry run of bench.py's rank-0 line assembly (the driver's contract) with a stand-in backend.

The build container has no GPU, so ``bench.worker`` cannot run here as shipped -- and a Python-level slip in the ~150 lines that
put the JSON line together would only show at round end on the GPU box. This test swaps the device backend for a stand-in that
answers every call bench.py makes with the ORACLE's results and made-up durations, and lets the planner entry points of the
drop-in leg run on the host-logic harness (oracle/planner_host.py). It checks the shape of the line, not any number in it:
the numbers are the GPU box's business (tests -m gpu, bench.py itself). Nothing here is product code.
"""
import argparse
import json
import types

import numpy as np
import pytest

import bench
from graphbasedlocaltrajectoryplanner_amd import _capi
from oracle.oracle_lib import OracleBackend
from oracle.planner_host import HostPlannerBackend       # bound to the real Planner class before the test patches the name
from oracle.fleet_host import HostFleetBackend
from graphbasedlocaltrajectoryplanner_amd import fleet as fleet_mod


class StandInFleet(object):
    """Fleet's surface as bench.py uses it, on the one-lane host build of the fleet state machine (oracle/fleet_host.py): the tape is kept in
    Python and replayed through the per-call entry points."""

    def __init__(self, hip, n_planners, **cfg):
        self.p = hip.fleet_host.planner(n_planners, **cfg)
        self.n, self.tape = n_planners, []

    def set_start(self, *a, **k):
        return self.p.set_start(*a, **k)

    def set_start_range(self, first, past_last, *a, **k):
        for q in range(first, past_last):
            r = self.p.set_start(q, *a, **k)
        return r

    def tape_append_groups(self, groups, ax_max_machines=((100.0, 5.0),)):
        self.tape.append((groups, ax_max_machines))

    def pack_groups(self, groups, ax_max_machines=((100.0, 5.0),)):
        return (groups, ax_max_machines), (groups, ax_max_machines), None

    def calc_paths_packed(self, packed):
        per = [g for c, g in packed[0] for _ in range(c)]
        self.p.calc_paths([g["prev_action"] for g in per], [g["t_now"] for g in per], [g["vehicles"] for g in per], [g["zone_gids"] for g in per])

    def calc_vel_profile_packed(self, packed):
        per = [g for c, g in packed[0] for _ in range(c)]
        g0 = per[0]
        self.p.calc_vel_profile([g["pos_est"] for g in per], [g["vel_est"] for g in per], vel_max=g0["vel_max"], gg_scale=[g["gg_scale"] for g in per],
                                local_gg=g0["local_gg"], ax_max_machines=packed[1], safety_d=[g["safety_d"] for g in per],
                                incl_emerg_traj=[g["incl_emerg_traj"] for g in per])

    def tape_run(self, first, count):
        for packed in self.tape[first:first + count]:
            self.calc_paths_packed(packed)
            self.calc_vel_profile_packed(packed)
        return 1.5 * count

    def trajectories(self, p):
        return self.p.trajectories(p)

    def close(self):
        self.p.close()


class StandInBackend(object):
    """HipBackend's surface as bench.py uses it; results from the oracle, durations invented."""

    def __init__(self, lattice, device=-1, persistent_tick=False):
        self.persistent_tick, self.n_single = persistent_tick, 0
        self.orc = OracleBackend(lattice)
        self.host = HostPlannerBackend(lattice)
        self.fleet_host = HostFleetBackend(lattice)
        self.caps = self.orc.caps
        self.calls = []
        self._resident = None
        import __graft_entry__ as ge
        self.lib_path = ge.HIP_LIB              # (the line reports the library's hashes and the digest of its batch path kernel)

    def paths_kernel_symbol(self, team_waves=1):
        return "_Z7k_pathsILi1E6PlanFxILi32ELi32ELi1EEE"

    def new_paths_result(self, n_scen):
        return self.orc.new_paths_result(n_scen)

    def persistent_stats(self):
        return {"enabled": int(self.persistent_tick), "resident": 0, "ticks": self.n_single, "launches": 1, "device_us_mean": 33.0, "device_us_last": 33.0,
                "idle_ms": 250.0}

    def tick_batch(self, batch, vel, result=None, vresult=None):
        self.calls.append("tick_batch")
        self.n_single += int(batch.n_scen == 1)
        return self.orc.tick_batch(batch, vel, result, vresult)

    def new_compact_trajectories(self, n_scen, max_rows=115, capacity_rows=None):
        return types.SimpleNamespace(struct=types.SimpleNamespace(total_rows=0), n_scen=n_scen, max_rows=max_rows)

    def tick_batch_compact(self, batch, vel, out):
        self.calls.append("tick_batch_compact")
        res, _ = self.orc.tick_batch(batch, vel)
        out.struct.total_rows = int(res.valid.sum()) * out.max_rows
        return out

    def batch_upload(self, batch, vel):
        self.calls.append("batch_upload")
        self._resident = (batch, vel)

    def batch_run(self, reps=1, timed=True):
        self.calls.append("batch_run")
        assert self._resident is not None
        return 0.9 * reps if timed else 0.0

    def batch_last_paths_ms(self):
        return 0.8

    def batch_run_profile(self, reps=10):
        return [0.8, 0.07, 0.15]

    def batch_download(self):
        return self.orc.tick_batch(*self._resident)

    def close(self):
        pass


def run_worker(monkeypatch, capsys, **over):
    import torch
    from graphbasedlocaltrajectoryplanner_amd import planner as planner_mod
    made = []

    def backend(lattice, device=-1, persistent_tick=False):
        made.append(StandInBackend(lattice, device, persistent_tick))
        return made[-1]

    monkeypatch.setattr(_capi, "HipBackend", backend)
    monkeypatch.setattr(planner_mod, "Planner", lambda hip, n_scen=1, **cfg: hip.host.planner(n_scen, **cfg))
    monkeypatch.setattr(fleet_mod, "Fleet", StandInFleet)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    args = dict(gpus=1, steps=3, warmup=1, batch=64, cpu_sample=8, latency_ticks=4, dropin_ticks=40, no_cpu=False,
                no_extra=False, exact_steps=True, workload="c2", fleet_planners=4, fleet_ticks=40, c3_batch=6, c5_ticks=2)
    args.update(over)
    bench.worker(argparse.Namespace(**args))
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    return json.loads(lines[0]), made[0]


def test_the_line_carries_the_drivers_contract(monkeypatch, capsys):
    out, hip = run_worker(monkeypatch, capsys)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(out[key], typ), key
    # no published reference number exists: vs_baseline is null; the ratio to BASELINE.json's stated target has its own name
    assert out["vs_baseline"] is None and out["vs_target"] == pytest.approx(out["value"] / 10000.0) and "target" in out["vs_target_basis"]
    assert out["unit"] == "ticks/s" and out["scaling"] == "weak" and out["dtype"] == "f64"        # (round 6: no fp32 operand anywhere on the path)
    assert len(out["per_rank_ms_per_step"]) == 1 and out["efficiency_vs_n1"] is None
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["timed_steps"] == 3
    assert "workload" in out["config"] and "model" not in out["config"]
    assert "ticks/s" in out["metric"] and out["value"] > 0 and np.isfinite(out["value"])
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 64.0) < 1e-6          # value = units of all ranks / timed region

    r = out["roofline"]
    # `contract_bound`: the roofline achieved / peak / frac are priced against (the bench contract's vocabulary); `bound`: what binds the
    # kernel -- the same word as binding.bound and limiter (round-5 review, item 9: the fields must agree)
    assert r["contract_bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBPS
    assert r["bound"] == "latency / " + r["binding"]["bound"] and r["limiter"].startswith(r["bound"])
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["kernel_ms"] == pytest.approx(0.8)                                # the live, in-region duration wins over the profile's
    assert r["traffic"] is None and r["traffic_frac"] is None                  # PMC summary is for the 32768-scenario grid only
    assert r["issue"] is None and r["binding"]["issue_frac"] is None
    assert r["frac_refline_per_position"] >= r["frac"]                         # (the round-3 model charged the reference line per position)
    # the re-based byte model never exceeds the survey's (which charged every window edge all its samples, every sweep its own edges)
    assert r["split_per_tick"]["mask"] <= r["survey_model_per_tick"]["mask"] and r["split_per_tick"]["sweep"] <= r["survey_model_per_tick"]["sweep"]
    assert r["mask_counts_per_tick"]["shell_edges"] <= r["mask_counts_per_tick"]["window_edges"]

    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "ticks/s" and c["value"] > 0 and "8 scenarios" in c["sample"]
    pd = out["parity_detail"]
    assert out["parity_checked"] is True and pd["scenarios"] == 8 and not pd["integer_mismatches"] and pd["paths"] >= 8
    assert set(pd["max_rel_err_by_quantity"]) >= {"x", "y", "psi", "kappa", "vx", "ax", "coeff_a1"}
    assert {"node_idx", "n_ties", "too_close", "nodes"} <= set(pd["integer_outputs_compared_bit_exact"])
    assert list(bench.sample_indices(32768, 2048)[[0, -1]]) == [0, 32767]      # the sample spans the whole batch

    l = out["latency_us"]
    assert l["ticks"] == 4 and l["p50"] > 0 and l["device_us"] == pytest.approx(900.0)
    assert l["dropin_ticks"] == 40 and l["dropin_keys_match_recording"] is True
    pt = l["persistent_tick"]                                                    # the same ticks through the resident kernel (own handle)
    assert pt["enabled"] is True and pt["ticks"] == 4 and pt["bit_identical"] is True and pt["device_us"] == pytest.approx(33.0) and pt["p99"] >= pt["p50"] > 0
    assert out["extra"]["pcie_inclusive"]["scenarios_per_call"] == 64 and out["extra"]["three_slot_paths_per_tick"] >= 1.0
    cl = out["extra"]["closed_loop"]
    assert cl["planners"] == 256 and cl["planner_ticks_per_s"] > 0 and cl["keys_match_recording"] is True
    cd = out["extra"]["closed_loop_device"]
    assert cd["planners"] == 4 and cd["ticks"] == 40 and cd["matches_recording"] is True and cd["planner_ticks_per_s"] == pytest.approx(4 * 40 / 0.06)
    assert cd["live_inputs_planner_ticks_per_s"] > 0 and "SAME" in cd["inputs"]
    cm = out["extra"]["closed_loop_device_mixed"]
    assert cm["planners"] == 4 and len(cm["groups"]) == 4 and cm["matches_recording"] is True and cm["planner_checks"] == 3 * 4 and len(cm["checked_at_ticks"]) == 3
    c3 = out["extra"]["c3"]
    assert c3["batch"] == 6 and c3["ticks_per_s"] > 0 and c3["parity_checked"] is True and 0 < c3["roofline_frac"] <= c3["roofline_frac_refline_per_position"]
    c4 = out["extra"]["c4"]                                                    # C4 as BASELINE states it (clipped to the dry run's batch)
    assert c4["one_gpu"]["scenarios"] == 64 and c4["shard"]["scenarios"] == 8 and c4["shard"]["resident_ticks_per_s"] > 0
    assert c4["one_gpu"]["pcie_us_per_call_p99"] >= c4["one_gpu"]["pcie_us_per_call_p50"] > 0 and "projection" in c4["projected_8gpu_ticks_per_s"]["basis"]
    c5 = out["extra"]["c5"]
    assert c5["horizon_300m"]["ticks"] == 2 and c5["horizon_100m"]["p99_us"] >= c5["horizon_100m"]["p50_us"] > 0
    assert 1.0 <= out["paths_per_tick"] <= 4.0
    # order of the device calls: resident inputs before any run, and the sample re-uploaded for the device-only latency
    assert hip.calls[0] == "batch_upload" and hip.calls.count("batch_upload") == 6


def test_traffic_is_reported_for_the_grid_it_was_measured_on(monkeypatch, capsys):
    import os
    with open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")) as fh:
        pmc = json.load(fh)
    n = int(pmc["grid_size"]) // 64
    assert bench.read_traffic(n, pmc.get("workload", "c2"))["hbm_bytes_per_launch"] == pmc["hbm_bytes_per_launch"]
    assert bench.read_traffic(n // 2, "c2") is None and bench.read_traffic(n // 2, "c3") is None
    c3p = os.path.join(bench.ROOT, "profiles", "pmc_traffic_c3.json")          # (C3 has its own files since round 5)
    if os.path.isfile(c3p):
        with open(c3p) as fh:
            c3 = json.load(fh)
        assert c3["workload"] == "c3" and bench.read_traffic(int(c3["grid_size"]) // 64, "c3")["hbm_bytes_per_launch"] == c3["hbm_bytes_per_launch"]
        assert bench.read_traffic(int(c3["grid_size"]) // 64, "c2") is None or pmc["grid_size"] == c3["grid_size"]


def test_counter_passes_are_tied_to_the_build(monkeypatch, capsys, tmp_path):
    """profiles/pmc_*.json carry the digest of the profiled library's path kernel (__graft_entry__.build_stamp); the line says whether it
    is the digest of the library the run drives: a stale file is flagged, not attached silently."""
    import __graft_entry__ as ge
    ge.build_hip()
    stamp = ge.build_stamp(ge.HIP_LIB, "_Z7k_pathsILi1E6PlanFxILi32ELi32ELi1EEE")
    assert stamp["isa_sha256"] and stamp["resources"]["vgpr_count"] <= 128 and stamp["isa_instructions"] > 1000
    monkeypatch.setattr(bench, "PROFILES_DIR", str(tmp_path))
    issue = {"kernel": "k_paths<1, PlanFx<32, 32, 1> >", "tag": "test", "workload": "c2", "grid_size": 64 * 64,
             "valu_insts_per_launch": 5.0e5, "salu_insts_per_launch": 3.0e5, "lds_insts_per_launch": 7.0e4,
             "valu_active_quad_cycles_per_launch": 5.2e5, "lanes_active_per_valu_inst": 45.0, "lds_active_quad_cycles_per_launch": 1.4e5,
             "wave_quad_cycles_per_launch": 3.0e6, "wait_any_quad_cycles_per_launch": 1.2e6}
    traffic = {"kernel": issue["kernel"], "tag": "test", "workload": "c2", "grid_size": 64 * 64, "hbm_bytes_per_launch": 1.4e6}
    for build, expect in ((dict(stamp, isa_sha256="0" * 64), False), (stamp, True), (None, None)):
        for name, d in (("pmc_issue.json", issue), ("pmc_traffic.json", traffic)):
            with open(tmp_path / name, "w") as fh:
                json.dump(dict(d, build=build) if build else d, fh)
        out, _ = run_worker(monkeypatch, capsys, no_cpu=True, no_extra=True, latency_ticks=0)
        r = out["roofline"]
        assert r["issue"]["build_matches"] is expect and r["traffic_build_matches"] is expect
        assert r["traffic"] == 1.4e6 and r["issue"]["valu_insts_per_scenario"] == pytest.approx(5.0e5 / 64)
        assert r["binding"]["wait_frac_of_wave_cycles"] == pytest.approx(0.4)
        lib = out["library"]
        assert lib["path"] == ge.HIP_LIB and lib["isa_sha256"] == stamp["isa_sha256"] and lib["lib_sha256"] == ge.file_sha256(ge.HIP_LIB)


def test_flags_that_drop_legs(monkeypatch, capsys):
    out, hip = run_worker(monkeypatch, capsys, no_cpu=True, no_extra=True, latency_ticks=0)
    assert "cpu_baseline" not in out and "parity_checked" not in out and out["extra"] == {}
    assert out["latency_us"]["p50"] is None and out["latency_us"]["dropin_p50"] is None and out["latency_us"]["ticks"] == 0
    assert "tick_batch" not in hip.calls


def test_refuses_to_run_without_a_gpu(monkeypatch):
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    with pytest.raises(SystemExit, match="no CPU fallback"):
        bench.worker(argparse.Namespace(gpus=1, steps=1, warmup=0, batch=8, workload="c2"))


def _rank_main(rank, port, out_dir, strong=False):
    """One rank of the two-rank rehearsal (spawned process: patches by hand, no pytest fixtures here)."""
    import contextlib
    import os
    import torch
    from graphbasedlocaltrajectoryplanner_amd import planner as planner_mod
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "LTPL_BENCH_SHARE_GPU": "1"})       # share: gloo + CPU tensors for the max
    _capi.HipBackend = lambda lattice, device=-1, persistent_tick=False: StandInBackend(lattice, device, persistent_tick)
    planner_mod.Planner = lambda hip, n_scen=1, **cfg: hip.host.planner(n_scen, **cfg)
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    args = argparse.Namespace(gpus=2, steps=2, warmup=1, batch=64, cpu_sample=8, latency_ticks=0, dropin_ticks=0, no_cpu=False,
                              no_extra=True, exact_steps=True, workload="c2", scaling="strong" if strong else "weak", batch_total=100)
    with open(os.path.join(out_dir, "rank%d.out" % rank), "w") as fh, contextlib.redirect_stdout(fh):
        bench.worker(args)


def test_two_ranks_print_one_line_with_the_whole_job_rate(tmp_path):
    """N > 1 as the driver launches it (one process per rank, env rendezvous on 127.0.0.1), rehearsed over gloo: only rank 0
    prints, n_gpus is the world size, and value counts the scenarios of BOTH shards over the max-over-ranks region."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_rank_main, args=(port, str(tmp_path)), nprocs=2, join=True)
    out0 = [l for l in (tmp_path / "rank0.out").read_text().splitlines() if l.startswith("{")]
    assert len(out0) == 1 and (tmp_path / "rank1.out").read_text().strip() == ""
    out = json.loads(out0[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "x2" in out["config"]["parallelism"]
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 128.0) < 1e-6
    assert out["cpu_baseline"]["cores"] == 1 and out["parity_checked"] is True


def test_two_ranks_strong_scaling_share_one_fixed_batch(tmp_path):
    """--scaling strong --batch-total T (BASELINE config C4: a fixed batch block-partitioned over the ranks): value counts T scenarios per
    step, every rank runs its block of the SAME batch, the line says "strong"."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_rank_main, args=(port, str(tmp_path), True), nprocs=2, join=True)
    out0 = [l for l in (tmp_path / "rank0.out").read_text().splitlines() if l.startswith("{")]
    assert len(out0) == 1 and (tmp_path / "rank1.out").read_text().strip() == ""
    out = json.loads(out0[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["batch_total"] == 100 and out["config"]["batch_per_gpu"] == 50
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 100.0) < 1e-6
    assert out["parity_checked"] is True
