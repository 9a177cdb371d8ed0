"""Multi-GPU batch sharding for the rollout.

Graphs in a batch never interact, so the path shards by contiguous batch slices with NO data-path
collective; one all-gather of the predicted states (RCCL over xGMI when the backend is "nccl") at the end
lets every rank evaluate costs / run the MPPI update redundantly, as the north-star specifies.
The reference itself is single-process ("replicas only").
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` samples for `rank`."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total), per


def gather_sharded(local, total, per, group=None):
    """All-gather per-rank shards (<= per rows each, zero-padded to `per`) into the full (total, ...) tensor."""
    world = dist.get_world_size(group)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    return out[:total]


def dynamics_sharded(dynamics_fn, state, action, *args, group=None, **kwargs):
    """Run `dynamics_fn(state, action_shard, *args)` on this rank's slice of the action samples and
    all-gather `state_seqs` / `action_seqs` so every rank returns the full-batch result."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dynamics_fn(state, action, *args, **kwargs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    total = action.shape[0]
    lo, hi, per = shard_bounds(total, rank, world)
    out = dynamics_fn(state, action[lo:hi], *args, **kwargs) if hi > lo else None
    if out is None:   # more ranks than samples: contribute an empty shard
        n_obj = state.shape[0]
        out = {"state_seqs": torch.zeros((0, action.shape[1], n_obj, 3), device=action.device),
               "action_seqs": torch.zeros((0, action.shape[1], 4), device=action.device)}
    return {k: gather_sharded(v, total, per, group) for k, v in out.items()}
