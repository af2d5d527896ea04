"""Multi-process path (world_size 2, gloo, CPU): block-sharded scenarios give bit-identical results to one rank.
The per-rank backend is the oracle's C library here (no GPU in the build container); on the GPU box the same
``tick_sharded`` runs over ``HipBackend`` inside bench.py."""
import os
import socket
import numpy as np
import pytest

from helpers import ROOT
from graphbasedlocaltrajectoryplanner_amd.sharding import shard_bounds, RESULT_FIELDS, VEL_FIELDS


def test_shard_bounds_partition():
    for n in (0, 1, 7, 8, 1024, 1031):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _inputs(lat, n, seed):
    from graphbasedlocaltrajectoryplanner_amd import _capi
    from graphbasedlocaltrajectoryplanner_amd.scenario_gen import c2_scenarios
    scen, vels = c2_scenarios(lat, n, seed=seed)
    rng = np.random.default_rng(seed + 5)
    vplan = rng.uniform(5.0, 60.0, n)
    pos = np.array([lat.node_pos[lat.layer_off[s['start_node'][0]] + s['start_node'][1]] for s in scen])
    return scen, vels, vplan, pos, _capi.VelParamSet(len_veh=lat.veh_length)


def _worker(rank, world, port, n, seed, out_path):
    import torch.distributed as dist
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from graphbasedlocaltrajectoryplanner_amd.lattice import Lattice
    from graphbasedlocaltrajectoryplanner_amd.sharding import tick_sharded
    from oracle.oracle_lib import OracleBackend
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        lat = Lattice.load(os.path.join(ROOT, "tests", "golden", "monteblanco_lattice.npz"))
        scen, vels, vplan, pos, params = _inputs(lat, n, seed)
        lo, hi, res, vres = tick_sharded(OracleBackend(lat), scen, [0.0, 0.5, 0.8], params, vplan, vplan, pos, vels,
                                         rank=rank, world=world, dist=dist, gather=True)
        if rank == 0:
            assert (lo, hi) == (0, n)
            np.savez(out_path, **{"r_" + k: v for k, v in res.items()}, **{"v_" + k: v for k, v in vres.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_equal_single_rank(tmp_path, monteblanco, oracle_backend):
    import torch.multiprocessing as mp
    from graphbasedlocaltrajectoryplanner_amd.sharding import tick_sharded
    n, seed = 37, 3                      # odd size: ragged blocks
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, n, seed, out_path), nprocs=2, join=True)
    scen, vels, vplan, pos, params = _inputs(monteblanco, n, seed)
    _, _, res, vres = tick_sharded(oracle_backend, scen, [0.0, 0.5, 0.8], params, vplan, vplan, pos, vels)
    with np.load(out_path) as z:
        for k in RESULT_FIELDS:
            assert np.array_equal(z["r_" + k], getattr(res, k)), k
        for k in VEL_FIELDS:
            assert np.array_equal(z["v_" + k], getattr(vres, k)), k
    assert int(res.valid.sum()) >= n
