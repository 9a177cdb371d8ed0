"""Multi-GPU batch sharding for the rollout.

Graphs in a batch never interact, so the path shards by contiguous batch slices with NO data-path
collective; one all-gather of the predicted states (RCCL over xGMI when the backend is "nccl") at the end
lets every rank evaluate costs / run the MPPI update redundantly, as the north-star specifies.
The reference itself is single-process ("replicas only").

Contract: `state` and `action` must be REPLICATED (identical on every rank) — each rank rolls out its slice of the
same action tensor and the gathered states are scored against the full one.  `replicate()` makes a tensor so
(broadcast from rank 0; `MPPIPlanner.sample` uses it for the sampled action sequences) and `assert_replicated()`
checks it.
"""
import collections

import torch
import torch.distributed as dist

_BUFFERS = collections.OrderedDict()
_BUFFERS_MAX = 8      # live shapes: (recv state_seqs, recv action_seqs, send pad) x the two batch sizes an MPPI loop alternates between


def _buffer(tag, shape, dtype, device):
    """Shape-keyed scratch tensors for the collective (send pad / receive buffer), allocated once per shape and kept in a small LRU: a planner
    that varies its sample count from call to call must not pile up one receive buffer per count (the bug class forward_dynamics._constants
    had).  A `copy=False` result is a view of such a buffer: valid until _BUFFERS_MAX other shapes have been gathered since."""
    key = (tag, tuple(shape), dtype, device.type, device.index)
    buf = _BUFFERS.get(key)
    if buf is None:
        while len(_BUFFERS) >= _BUFFERS_MAX:
            _BUFFERS.popitem(last=False)
        buf = torch.empty(shape, dtype=dtype, device=device)
        _BUFFERS[key] = buf
    else:
        _BUFFERS.move_to_end(key)
    return buf


def active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def shard_bounds(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` samples for `rank`."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total), per


def replicate(t, group=None, src=0):
    """Make `t` identical on every rank (broadcast from `src`); no-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.contiguous()
        dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return t


def assert_replicated(t, what="tensor", group=None):
    """Raise unless `t` is bit-identical on all ranks (two tiny all-reduces + one host read: for set-up / tests, not
    for the timed loop)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    probe = t.detach().to(torch.float64)
    sig = torch.stack([probe.sum(), (probe * probe).sum(), probe.flatten()[::7].sum()])
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"adaptigraph_amd.dist: `{what}` differs across ranks; dynamics_sharded needs replicated "
                           "inputs (use dist.replicate() or seed every rank identically)")


def gather_sharded(local, total, per, group=None, copy=True, tag=""):
    """All-gather per-rank shards (<= per rows each) into the full (total, ...) tensor.

    Shards are contiguous and balanced (shard_bounds), so every rank before the last non-empty one is full and the
    first `total` rows of the rank-major receive buffer are exactly the valid rows.  A full shard is sent in place
    (no pad copy); a short one is staged into a cached `per`-row send buffer whose tail is never read.  Send and
    receive buffers are allocated once per shape.  `copy=False` returns a view of the cached receive buffer (valid
    until the next gather of the same shape)."""
    world = dist.get_world_size(group)
    tail = tuple(local.shape[1:])
    out = _buffer("recv" + tag, (world * per,) + tail, local.dtype, local.device)
    if local.shape[0] == per and local.is_contiguous():
        send = local
    else:
        send = _buffer("send", (per,) + tail, local.dtype, local.device)
        send[: local.shape[0]] = local
    dist.all_gather_into_tensor(out, send, group=group)
    res = out[:total]
    return res.clone() if copy else res


def dynamics_sharded(dynamics_fn, state, action, *args, group=None, timing=None, copy=True, **kwargs):
    """Run `dynamics_fn(state, action_shard, *args)` on this rank's slice of the (replicated) action samples and
    all-gather `state_seqs` / `action_seqs` so every rank returns the full-batch result.

    `copy=False`: the returned tensors are views of the cached receive buffers (no extra device copy of the gathered states:
    12 MB per call at BASELINE configs[4]); they stay valid until the next call with the same shapes — what a planner that scores
    the states right away (mpc.py, bench.py) wants.

    `timing`, if given, is a dict of lists: a (start, end) pair of CUDA events is appended under "rollout" and
    "gather" for the local rollout and the collective of this call (resolve with `elapsed_ms` after a sync)."""
    if not active(group):
        return dynamics_fn(state, action, *args, **kwargs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    total = action.shape[0]
    lo, hi, per = shard_bounds(total, rank, world)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timing is not None else None
    if ev:
        ev[0].record()
    out = dynamics_fn(state, action[lo:hi], *args, **kwargs) if hi > lo else None
    if out is None:   # more ranks than samples: contribute an empty shard
        n_obj = state.shape[0]
        out = {"state_seqs": torch.zeros((0, action.shape[1], n_obj, 3), device=action.device),
               "action_seqs": torch.zeros((0, action.shape[1], 4), device=action.device)}
    if ev:
        ev[1].record()
    res = {k: gather_sharded(v, total, per, group, copy=copy, tag=":" + k) for k, v in out.items()}
    if ev:
        ev[2].record()
        timing.setdefault("rollout", []).append((ev[0], ev[1]))
        timing.setdefault("gather", []).append((ev[1], ev[2]))
    return res


def elapsed_ms(pairs):
    """Sum of the elapsed times of (start, end) CUDA event pairs (call after torch.cuda.synchronize())."""
    return float(sum(a.elapsed_time(b) for a, b in pairs))
