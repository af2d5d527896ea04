"""
Scenario sharding for the batched-throughput path (BASELINE config C4).

Planning problems of different scenarios share only the read-only lattice, so the batch is block-partitioned over the
ranks of one node (one process per GPU, lattice replicated by one ``ltpl_create`` per rank) and there is NO collective
on the data path. ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) is only
used to (optionally) gather the per-rank result slabs on rank 0 and for the benchmark's barrier / max-over-ranks.
"""
import numpy as np

from . import _capi


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank ``rank``; the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("invalid rank / world size")
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


RESULT_FIELDS = ("end_layer", "closest_obj_index", "closest_obj_node", "n_actions", "action_id", "valid", "reduced",
                 "goal_layer", "n_nodes", "n_pts", "n_ties", "nodes", "node_idx", "coeff", "path_param")
VEL_FIELDS = ("vx", "ax", "vel_bound", "too_close")


def tick_sharded(backend, scenarios, w_last_edges, vel_params, vel_plan, vel_est, pos_est, veh_vels, rank=0, world=1,
                 dist=None, gather=True, **tick_kw):
    """
    Run the fused tick on this rank's block of ``scenarios`` (list of scenario dicts, identical on every rank; the
    per-scenario arrays ``vel_plan / vel_est / pos_est`` and the list ``veh_vels`` of per-scenario vehicle-speed arrays
    likewise). Returns (lo, hi, result, vel_result) of the local block, or -- with ``gather`` and a process group -- on
    rank 0 the dicts of concatenated arrays for the whole batch (None on the other ranks).
    """
    n = len(scenarios)
    lo, hi = shard_bounds(n, rank, world)
    local = None
    if hi > lo:
        batch = _capi.PathsBatch(scenarios[lo:hi], w_last_edges=w_last_edges)
        vv = [np.asarray(v, dtype=np.float64) for v in veh_vels[lo:hi]]
        vel = _capi.TickVelBatch(vel_params, hi - lo, np.asarray(vel_plan)[lo:hi], np.asarray(vel_est)[lo:hi],
                                 np.asarray(pos_est)[lo:hi], np.concatenate(vv) if vv and sum(len(v) for v in vv) else np.zeros(0),
                                 **tick_kw)
        local = backend.tick_batch(batch, vel)
    if not gather or dist is None or world == 1:
        return lo, hi, (local[0] if local else None), (local[1] if local else None)
    payload = None
    if local is not None:
        payload = ({k: getattr(local[0], k) for k in RESULT_FIELDS}, {k: getattr(local[1], k) for k in VEL_FIELDS})
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((lo, hi, payload), gathered, dst=0)
    if rank != 0:
        return lo, hi, None, None
    gathered = sorted([g for g in gathered if g[2] is not None], key=lambda g: g[0])
    res = {k: np.concatenate([g[2][0][k] for g in gathered], axis=0) for k in RESULT_FIELDS}
    vres = {k: np.concatenate([g[2][1][k] for g in gathered], axis=0) for k in VEL_FIELDS}
    return 0, n, res, vres


def fleet_shard(n_vehicles: int, rank: int, world: int):
    """Planners [lo, hi) of a fleet of ``n_vehicles`` that rank ``rank`` owns (``fleet.Fleet``, ltpl_fleet_*): vehicles share nothing but
    the read-only lattice, so a multi-GPU fleet is one fleet per rank on its block of vehicles -- weak scaling, no collective on the data
    path, exactly as the scenario batches above. Inputs of a tick are sliced with the same bounds (CSR offsets re-based by the caller)."""
    return shard_bounds(n_vehicles, rank, world)
