"""Multi-process fleet (world size 2, gloo, CPU): every rank runs ITS block of vehicles as its own fleet (one-lane host build of the fleet
state machine over the oracle's arithmetic here; ltpl_fleet_* on one GPU per rank on the GPU box), no collective on the data path; the
gathered trajectories equal those of one fleet that holds all vehicles."""
import os
import socket

import numpy as np

from helpers import ROOT

N_VEH, N_TICKS, NAMES = 5, 70, ("c1", "zonewall")


def _run(lat, lo, hi):
    """Vehicles lo .. hi - 1 (vehicle v replays recording NAMES[v % 2]) for N_TICKS ticks; returns {vehicle: [(key, array), ...]}."""
    import planner_replay as pr
    from oracle.fleet_host import HostFleetBackend
    recs = [pr.load_ticks(nm) for nm in NAMES]
    fleet = HostFleetBackend(lat).planner(hi - lo)
    for v in range(lo, hi):
        st = recs[v % 2][0]['start']
        fleet.set_start(v - lo, st['pos'], st['heading'], st['vel'], st['max_heading_offset'])
    for k in range(N_TICKS):
        ts = [recs[v % 2][k] for v in range(lo, hi)]
        va = ts[0]['vel_args']
        fleet.calc_paths([t['action_id_sel'] for t in ts], [t['t'] for t in ts], [pr.vehicles_of_tick(t) for t in ts],
                         [pr.zone_gids_of_tick(lat, t) for t in ts])
        fleet.calc_vel_profile([t['pos_est'] for t in ts], [t['vel_args']['vel_est'] for t in ts], vel_max=va['vel_max'], gg_scale=va['gg_scale'],
                               local_gg=tuple(va['local_gg']), ax_max_machines=va['ax_max_machines'], safety_d=va['safety_d'],
                               incl_emerg_traj=[t['vel_args']['incl_emerg_traj'] for t in ts])
    return {v: [(k, a[0]) for k, a in fleet.trajectories(v - lo)[0].items()] for v in range(lo, hi)}


def _worker(rank, world, port, out_path):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    from graphbasedlocaltrajectoryplanner_amd.sharding import fleet_shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
        lo, hi = fleet_shard(N_VEH, rank, world)
        mine = _run(lat, lo, hi)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)           # (results only: the ticks themselves needed no communication)
        if rank == 0:
            merged = {}
            for g in gathered:
                merged.update(g)
            np.savez(out_path, **{"v%d_%d_%s" % (v, i, k): a for v, lst in merged.items() for i, (k, a) in enumerate(lst)})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_fleet_equals_one_fleet(tmp_path, monteblanco):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "fleet.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    whole = _run(monteblanco, 0, N_VEH)
    with np.load(out) as z:
        got = {k: z[k] for k in z.files}
    assert len(got) == sum(len(lst) for lst in whole.values())
    for v, lst in whole.items():
        for i, (k, a) in enumerate(lst):
            assert np.array_equal(got["v%d_%d_%s" % (v, i, k)], a), (v, k)
