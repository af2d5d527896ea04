#!/usr/bin/env python
"""
bench.py -- planning ticks/s of the fused MI355X tick (seam 1 + per-primitive velocity stage) on the C2 workload.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one launch of the fused tick kernel over one batch of independent C2 scenarios (Monteblanco lattice, 4
action primitives, 8 dynamic opponents with a 0.2 s prediction each = 16 obstacle positions, sample zone) whose
inputs are already resident in HBM (ltpl_batch_upload). Every rank owns one GPU and its own shard of scenarios (weak
scaling, no data-path collective: scenarios are independent); torch.distributed (RCCL) is only used for
the barrier and the max-over-ranks of the timed region.

The JSON line also carries
  roofline      algorithmic bytes per launch (graphbasedlocaltrajectoryplanner_amd/roofline.py) divided by
                the DOMINANT kernel's (path kernel: mask + sweeps + spline) average duration measured with HIP events on
                the library's own stream, against the 8 TB/s HBM peak of MI355X; `traffic` = HBM bytes per launch of that
                kernel from the rocprofv3 PMC passes, if profiles/ holds them; `pipeline_ms` lists all kernels of a step
  cpu_baseline  the oracle's plain-C restatement (kind "port", 1 core) timed on a bounded sample of the same scenarios
  latency_us    p50 / p99 of single-scenario ticks through ltpl_tick_batch including all host marshalling and PCIe
"""
import argparse
import json
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


def make_batch(lat, n, seed, workload="c2"):
    if workload == "c3":
        from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import scattered_obstacle_scenarios
        scen, vels = scattered_obstacle_scenarios(lat, n, n_obj=32, seed=seed)
    else:
        scen, vels = c2_scenarios(lat, n, seed=seed)
    rng = np.random.default_rng(seed + 77)
    params = _capi.VelParamSet(len_veh=lat.veh_length)      # Graph_LTPL.calc_vel_profile defaults (Graph_LTPL.py:347-351)
    vplan = rng.uniform(5.0, 60.0, n)
    pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])       # params/ltpl_config_online.ini:71
    vel = _capi.TickVelBatch(params, n, vplan, vplan, pos, np.concatenate(vels))
    return scen, batch, vel


def cpu_baseline(lat, scen_batch, vel, n_sample):
    """Oracle (plain-C restatement, oracle/ltpl_oracle.c) on the first n_sample scenarios of rank 0's shard, 1 core."""
    from oracle.oracle_lib import OracleBackend
    orc = OracleBackend(lat)
    scen = scen_batch[:n_sample]
    batch = _capi.PathsBatch(scen, w_last_edges=[0.0, 0.5, 0.8])
    n_veh = int(batch.veh_off[-1])
    v = _capi.TickVelBatch(vel.params, len(scen), vel.vel_plan[:n_sample], vel.vel_est[:n_sample],
                           np.column_stack((vel.pos_x[:n_sample], vel.pos_y[:n_sample])), vel.veh_vel[:n_veh])
    orc.tick_batch(batch, v)                                    # warm caches
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.tick_batch(batch, v)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 200:
            break
    return {"value": len(scen) * reps / el, "unit": "ticks/s", "cores": 1, "kind": "port",
            "sample": "%d scenarios of the same workload x %d passes through oracle_tick_batch (plain C, -O2, single thread)"
                      % (len(scen), reps)}


def read_traffic(batch, workload):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary (profiles/pmc_traffic.json), if it
    was collected for this workload and batch size (grid = 64 threads x batch); otherwise null."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(p):
        try:
            with open(p) as fh:
                d = json.load(fh)
            if d.get("workload", "c2") == workload and int(d.get("grid_size", 0)) == 64 * batch:
                return d.get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32768, help="scenarios per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--latency-ticks", type=int, default=2000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=("c2", "c3"), default="c2",
                    help="c2 = BASELINE config the metric is quoted on (default); c3 = synthetic 10k-node / 98k-edge lattice "
                         "with 32 obstacles per scenario (HBM-roofline run, reported separately under profiles/)")
    args = ap.parse_args()

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
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))

    if args.workload == "c3":
        from graphbasedlocaltrajectoryplanner_amd.synthetic_lattice import c3_lattice
        lat = c3_lattice()
    else:
        lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
    hip = _capi.HipBackend(lat, device=dev_index)
    scen, batch, vel = make_batch(lat, args.batch, seed=1 + rank, workload=args.workload)
    hip.batch_upload(batch, vel)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed)
    for _ in range(args.warmup):
        hip.batch_run(reps=1, timed=False)
    barrier()
    t0 = time.perf_counter()
    ms_kernel = hip.batch_run(reps=args.steps, timed=True)      # HIP events on the library's stream + wait
    paths_ms_live = hip.batch_last_paths_ms()                    # path kernel inside the timed region (events per launch)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, vres = hip.batch_download()

    # per-kernel durations of the pipeline (HIP events between the launches on the library's stream), outside the timed region
    prof_ms = hip.batch_run_profile(reps=min(args.steps, 20))
    if rank == 0:
        ab = algorithmic_bytes(lat, batch, res)
        kern_ms = ms_kernel / args.steps
        n_paths = int(res.valid.sum())
        # dominant kernel = the path kernel (mask + sweeps + spline); its algorithmic bytes exclude the velocity stage
        ab_paths = ab["mask"] + ab["sweep"] + ab["path"]
        dom_ms = paths_ms_live if paths_ms_live > 0.0 else prof_ms[0]
        achieved = ab_paths / (dom_ms * 1e-3) / 1e9
        # single-scenario latency through the synchronous C call (host marshalling + H2D + kernel + D2H)
        lat_us = []
        one_res, one_vres = hip.new_paths_result(1), _capi.TickVelResult(1, hip.caps.max_path_pts)
        singles = []
        for i in range(64):
            b1 = _capi.PathsBatch([scen[i]], w_last_edges=[0.0, 0.5, 0.8])
            v1 = _capi.TickVelBatch(vel.params, 1, vel.vel_plan[i:i + 1], vel.vel_est[i:i + 1],
                                    np.array([[vel.pos_x[i], vel.pos_y[i]]]),
                                    vel.veh_vel[batch.veh_off[i]:batch.veh_off[i + 1]])
            singles.append((b1, v1))
        for i in range((100 + args.latency_ticks) if args.latency_ticks > 0 else 0):
            b1, v1 = singles[i % 64]
            t1 = time.perf_counter()
            hip.tick_batch(b1, v1, one_res, one_vres)
            if i >= 100:
                lat_us.append((time.perf_counter() - t1) * 1e6)
        lat_us = np.array(lat_us)
        # device-only time of one single-scenario tick (latency method): the same fused kernel launched
        # back to back on a device-resident scenario, HIP events on the library's stream
        device_us = None
        if args.latency_ticks > 0:
            hip.batch_upload(singles[0][0], singles[0][1])
            hip.batch_run(reps=20, timed=False)
            device_us = hip.batch_run(reps=200, timed=True) / 200 * 1e3
        out = {
            "metric": "planning ticks/s (all action primitives), " + ("Monteblanco lattice" if args.workload == "c2" else "synthetic C3 lattice"),
            "value": world * args.batch * args.steps / elapsed,
            "unit": "ticks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (("C2: Monteblanco lattice (%d layers / %d nodes / %d edges), 4 action primitives, 8 dynamic "
                                     "opponents (16 obstacle positions), sample zone" if args.workload == "c2" else
                                     "C3: synthetic oval lattice (%d layers / %d nodes / %d edges), 4 action primitives, 32 static "
                                     "obstacles (64 obstacle positions)") % (lat.num_layers, lat.num_nodes, lat.num_edges))
                                   + "; %d independent scenarios per GPU per step, tick pipeline (paths + velocity)" % args.batch,
                       "batch_per_gpu": args.batch, "parallelism": "scenario-sharded x%d (no collective)" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": read_traffic(args.batch, args.workload),
                         "kernel": "k_paths<1>", "kernel_ms": dom_ms,
                         "kernel_ms_not_overlapped": prof_ms[0],
                         "algorithmic_bytes_per_launch": ab_paths,
                         "algorithmic_bytes_per_tick": ab["total"] / args.batch,
                         "split_per_tick": {k: ab[k] / args.batch for k in ("mask", "sweep", "path", "vel")},
                         "pipeline_ms": {"k_paths": prof_ms[0], "k_follow_prep": prof_ms[1], "k_vel_lanes": prof_ms[2],
                                         "all_kernels_back_to_back": kern_ms},
                         "whole_tick_achieved": ab["total"] / (kern_ms * 1e-3) / 1e9},
            "latency_us": {"p50": float(np.percentile(lat_us, 50)) if lat_us.size else None,
                           "p99": float(np.percentile(lat_us, 99)) if lat_us.size else None,
                           "mean": float(lat_us.mean()) if lat_us.size else None, "ticks": int(lat_us.size),
                           "device_us": device_us,
                           "what": "one scenario per ltpl_tick_batch call, host wall time incl. marshalling + PCIe; device_us = the tick kernel alone (HIP events, back-to-back launches)"},
            "paths_per_tick": n_paths / args.batch,
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(lat, scen, vel, min(args.cpu_sample, args.batch))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
