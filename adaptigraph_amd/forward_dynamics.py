"""Batched multi-step rollout drivers on the HIP engine — drop-in for src/planning/forward_dynamics.py.

`dynamics(state, action, model, device, ppm_optimizer, physics_param=None)` (:11-205) and
`dynamics_masked(state_init, state_mask, action, model, device, ppm_optimizer, physics_param=None)` (:208-399)
keep the reference signatures and return dicts.  The per-sample set-up (tool key-points from the decoded
action) is a handful of tiny device tensor ops; the inner loop — edges -> GNN forward -> record-on-repeat ->
tool advance -> history shift -> edge rebuild — is one `ag_rollout` call with no host synchronisation
(the reference syncs three times per step: truncate_graph, n_rels.max().item(), pad_torch).
"""
import collections
import ctypes
import os

import torch

from . import _lib
from .graph import threshold_sq, workspace, _stream_ptr
from .plan_utils import decode_action


def _u8(t):
    return t.contiguous().view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8).contiguous()


def rollout(model, state0, delta, attrs, p_instance, phys, mask, tool_mask, thr_sq, repeat, n_steps, topk,
            connect_tools_all, max_tools, height_mode=_lib.AG_HEIGHT_MIN, obj_mask=None, gripper_raise=0.0,
            return_state=False, out=None):
    """Native inner loop (forward_dynamics.py:156-197).  All tensors on the model's GPU.
    state0 (B,H,N,3), delta (B,N,3), attrs (B,N,2), p_instance (B,n_p,I), phys (B,P), mask/tool_mask (B,N) bool,
    thr_sq (B,), repeat (B,) int32 -> out_seq (B,n_p,3): prediction of step repeat[b] (zeros if never reached).
    `out`: an already ZEROED contiguous fp32 (B,n_p,3) tensor to write into instead of a fresh one."""
    L = _lib.lib()
    dev = state0.device
    B, H, N, _ = state0.shape
    n_p, n_inst = p_instance.shape[1], p_instance.shape[2]
    prm = _lib.RolloutParams(B, N, n_p, n_inst, int(topk), 1 if connect_tools_all else 0, int(max_tools), int(n_steps),
                             int(height_mode), float(gripper_raise))
    if out is not None:
        assert out.shape == (B, n_p, 3) and out.dtype == torch.float32 and out.is_contiguous() and out.device == dev
    out_seq = out if out is not None else torch.zeros((B, n_p, 3), dtype=torch.float32, device=dev)
    state_final = torch.empty_like(state0, dtype=torch.float32) if return_state else None
    h = model.handle(dev)
    ws = workspace(dev, L.ag_rollout_workspace_bytes_for(h, ctypes.byref(prm)))      # exact for the model's current mode (q16 table: half the largest buffer)
    state0 = state0.contiguous().float()
    delta = delta.contiguous().float()
    attrs = attrs.contiguous().float()
    p_instance = p_instance.contiguous().float()
    phys = phys.to(dev, torch.float32).contiguous()
    mask_u8, tool_u8 = _u8(mask), _u8(tool_mask)
    obj_u8 = _u8(obj_mask) if obj_mask is not None else None
    repeat = repeat.to(dev, torch.int32).contiguous()
    thr_sq = thr_sq.contiguous()
    with torch.cuda.device(dev):
        rc = L.ag_rollout(h, ctypes.byref(prm), state0.data_ptr(), delta.data_ptr(), attrs.data_ptr(),
                          p_instance.data_ptr(), phys.data_ptr(), mask_u8.data_ptr(), tool_u8.data_ptr(),
                          obj_u8.data_ptr() if obj_u8 is not None else None, thr_sq.data_ptr(), repeat.data_ptr(),
                          out_seq.data_ptr(), state_final.data_ptr() if return_state else None, ws.data_ptr(),
                          ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "ag_rollout")
    return (out_seq, state_final) if return_state else out_seq


def _place_tool(task, decoded, theta, y, device):
    """Tool key-points and per-step delta from a decoded action (forward_dynamics.py:42-81 / :237-276)."""
    bsz = decoded.shape[0]
    pts = task["pusher_points"]
    ratio = task["sim_real_ratio"]
    n_t = len(pts)
    eef = torch.zeros((bsz, n_t, 3), device=device)
    dlt = torch.zeros((bsz, n_t, 3), device=device)
    dlt[:, :, 0] = (decoded[:, 2] - decoded[:, 0]).unsqueeze(1)
    dlt[:, :, 2] = (decoded[:, 3] - decoded[:, 1]).unsqueeze(1)
    eef[:, :, 1] = y[:, None]
    if n_t == 1:
        eef[:, 0, 0] = decoded[:, 0]
        eef[:, 0, 2] = decoded[:, 1]
    elif n_t == 5:
        eef[:, 0, 0] = decoded[:, 0]
        eef[:, 0, 2] = decoded[:, 1]
        for i in range(1, 5):
            off = float(pts[i][1]) * ratio
            eef[:, i, 0] = decoded[:, 0] + off * torch.sin(theta)
            eef[:, i, 2] = decoded[:, 1] - off * torch.cos(theta)
    else:
        raise NotImplementedError("pusher not implemented")
    raise_by = 0.01 * ratio if task["gripper_enable"] else 0.0
    if raise_by:
        eef[:, :, 1] += raise_by
    return eef, dlt, raise_by


def _place_tool_lean(task, decoded, theta, y, device, zero=None):
    """_place_tool with the same values in fewer launches (dynamics() issues this set-up once per look-ahead step in front of every rollout:
    a dozen 4-us fills and slice assignments are 0.4 % of a 256 x 10 pass and more of an MPPI chunk): the key-point table is assembled by ONE
    stack, the per-step motion by one subtraction + one stack; every element is produced by the same fp32 operation as in _place_tool
    (tests/test_host_logic.py compares the two bit for bit)."""
    pts = task["pusher_points"]
    ratio = task["sim_real_ratio"]
    n_t = len(pts)
    raise_by = 0.01 * ratio if task["gripper_enable"] else 0.0
    d = decoded[:, 2:4] - decoded[:, 0:2]
    dlt = torch.stack([d[:, 0], torch.zeros_like(d[:, 0]) if zero is None else zero, d[:, 1]], dim=-1)[:, None].expand(-1, n_t, -1)      # (`zero`: a cached (bsz,) zero column)
    yy = y + raise_by if raise_by else y
    if n_t == 1:
        eef = torch.stack([decoded[:, 0], yy, decoded[:, 1]], dim=-1)[:, None]
    elif n_t == 5:
        key = ("pusher_off", tuple(float(pts[i][1]) * ratio for i in range(1, 5)), str(device))      # (a host list -> device tensor is a synchronising copy: once per pusher)
        off = _OFFSETS.get(key)
        if off is None:
            off = _OFFSETS[key] = torch.tensor([0.0] + list(key[1]), device=device)      # (key-point 0: x + 0 * sin is x + 0)
        sn, cs = torch.sin(theta), torch.cos(theta)
        ex = decoded[:, 0:1] + off[None] * sn[:, None]
        ez = decoded[:, 1:2] - off[None] * cs[:, None]
        ex[:, 0] = decoded[:, 0]         # key-point 0 is the decoded start itself (x + 0 * sin(theta) would turn a -0.0 into +0.0 and an inf * 0 into NaN)
        ez[:, 0] = decoded[:, 1]
        eef = torch.stack([ex, yy[:, None].expand(-1, 5), ez], dim=-1)
    else:
        raise NotImplementedError("pusher not implemented")
    return eef, dlt, raise_by


def _physics(ppm_optimizer, physics_param, bsz, device):
    physics_param = ppm_optimizer.physics_param if physics_param is None else physics_param
    material = ppm_optimizer.material
    dims = ppm_optimizer.material_dims
    assert len(dims) == 1 and material in dims, "Only support single material."
    p = physics_param[material].to(device, torch.float32)
    if p.dim() == 2:      # extension over the reference: one parameter vector PER SAMPLE (batched sys-id sweeps)
        assert p.shape[0] == bsz, f"per-sample physics_param needs {bsz} rows, got {p.shape[0]}"
        return p.contiguous()
    return p[None].repeat(bsz, 1)


_CONST = collections.OrderedDict()      # per (batch, particles, tools, instances, device): the call-invariant inputs of dynamics(), LRU of _CONST_MAX
_CONST_MAX = 4                          # entries (an MPPI loop alternates between its chunk size and the bsz = 1 best-sample rollout: 2 live keys)
_CONST_MAX_BYTES = 256 << 20            # and device bytes (20 000 samples x 200 particles: 66 MB per entry)
_PINNED = {}
_OFFSETS = {}                           # per (pusher geometry, device): key-point offsets of _place_tool_lean


def _pinned(n, dtype, device):
    """One pinned host buffer per (length, dtype, device): streams of different devices never share a staging buffer."""
    key = (n, dtype, str(device))
    t = _PINNED.get(key)
    if t is None:
        t = _PINNED[key] = torch.empty(n, dtype=dtype, pin_memory=True)
    return t


def _strict_status(model, device):
    """AG_STRICT_STATUS=1: read the model's numeric status at the END of the call that may have raised it (one more host round trip per call)
    and raise instead of warning; by default the status of call k is seen by call k + 1's read (sync-free), or by an explicit
    model.take_status()."""
    if os.environ.get("AG_STRICT_STATUS", "0") == "1":
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            flags = model.take_status(device)
        if flags:
            raise FloatingPointError(f"adaptigraph_amd: this rollout left the range of its arithmetic (ag_model_status = {flags}): non-finite "
                                     "inputs, or an fp16 activation of the 'fast' edge stack beyond 65504 — use model.set_option('precision', 1)")


def _constants(bsz, n_obj, n_t, max_n, device):
    """attrs / p_instance / mask / tool_mask of forward_dynamics.py:83-123 depend only on the shapes: built once per shape (a dozen tiny
    launches per call otherwise, issued while the GPU idles behind the call's one host sync).  A small LRU bounded by entries AND bytes, so
    varying MPPI chunk sizes (20 000 samples, chunks, the bsz = 1 best-sample rollout) cannot pile up device memory.
    READ-ONLY: the tensors are shared by every later call with the same shapes — callers must not write into them."""
    key = (bsz, n_obj, n_t, max_n, str(device))
    c = _CONST.get(key)
    if c is not None:
        _CONST.move_to_end(key)
        return c
    N = n_obj + n_t
    attrs = torch.zeros((bsz, N, 2), device=device)
    attrs[:, :n_obj, 0] = 1.0
    attrs[:, n_obj:, 1] = 1.0
    p_instance = torch.zeros((bsz, n_obj, max_n), device=device)
    p_instance[:, :, 0] = 1.0
    mask = torch.ones((bsz, N), dtype=torch.bool, device=device)
    tool_mask = torch.zeros((bsz, N), dtype=torch.bool, device=device)
    tool_mask[:, n_obj:] = True
    zeros = torch.zeros((1, n_obj, 3), device=device)        # expanded per call: the object particles' per-step motion (forward_dynamics.py:116-123) and a zero per sample
    c = (attrs, p_instance, mask, tool_mask, zeros)
    nbytes = lambda e: sum(t.numel() * t.element_size() for t in e)      # noqa: E731
    while _CONST and (len(_CONST) >= _CONST_MAX or sum(map(nbytes, _CONST.values())) + nbytes(c) > _CONST_MAX_BYTES):
        _CONST.popitem(last=False)
    _CONST[key] = c
    return c


@torch.no_grad()
def dynamics(state, action, model, device, ppm_optimizer, physics_param=None):
    task = ppm_optimizer.task_config
    n_his, push_length = task["n_his"], task["push_length"]
    state = state.to(device, torch.float32)
    action = action.to(device, torch.float32)
    bsz, n_look = action.shape[0], action.shape[1]
    decoded, repeat = decode_action(action, push_length=push_length)
    n_obj, n_t = state.shape[0], ppm_optimizer.eef_num
    N = n_obj + n_t

    attrs, p_instance, mask, tool_mask, obj_still = _constants(bsz, n_obj, n_t, task["max_n"], device)
    obj_still = obj_still.expand(bsz, n_obj, 3)
    phys = _physics(ppm_optimizer, physics_param, bsz, device)
    thr = threshold_sq(ppm_optimizer.adj_thresh, bsz, torch.device(device), _lib.AG_VARIANT_BATCH)
    rep_max = repeat.max(dim=0).values
    rep_cols = repeat.to(torch.int32).t().contiguous()      # (n_look, B): each look-ahead step's column contiguous, made before the sync
    seq = torch.zeros((bsz, n_look, n_obj, 3), device=device)

    def prepare(li, obj, shared=False):        # tool key-points, history frames and per-step tool motion of look-ahead step li
        # (`shared`: every sample starts from the same cloud — its height is one reduction over n_obj values, not bsz of them: 28 -> 5 us at 256 x 1 000)
        y = obj[0, :, 1].min().expand(bsz) if shared else obj[:, :, 1].min(dim=1).values
        eef, dlt, raise_by = _place_tool_lean(task, decoded[:, li], action[:, li, 2], y, device, zero=obj_still[:, 0, 0])
        state0 = torch.cat([obj[:, None].expand(bsz, n_his, n_obj, 3), eef[:, None].expand(bsz, n_his, n_t, 3)], dim=2)
        delta = torch.cat([obj_still, dlt], dim=1)
        return state0, delta, raise_by

    # everything the first look-ahead step needs is enqueued BEFORE the call's one host sync, so that after it only the rollout's own launches
    # stand between the host and a busy GPU
    ready = prepare(0, state[None].expand(bsz, n_obj, 3), shared=True)
    # ONE host sync per call (reference: one per look-ahead + 3 per step): the step counts travel to pinned memory behind the set-up, and the read of
    # the model's deferred numeric status — which synchronises the stream — completes both (one host round trip, ~75 us each on these boxes)
    rep_host = _pinned(n_look, rep_max.dtype, device)
    rep_host.copy_(rep_max, non_blocking=True)
    copied = torch.cuda.Event()
    copied.record(torch.cuda.current_stream(torch.device(device)))
    model.take_status(device)
    copied.synchronize()      # normally already complete (take_status synchronises the same stream); never rely on a subclass / stub doing so
    max_steps = rep_host.tolist()
    for li in range(n_look):
        if li > 0:
            ready = prepare(li, seq[:, li - 1])
        state0, delta, raise_by = ready
        direct = seq[:, 0] if n_look == 1 else None     # a single look-ahead step: the library writes the result tensor itself
        res = rollout(model, state0, delta, attrs, p_instance, phys, mask, tool_mask, thr, rep_cols[li],
                      max_steps[li], task["topk"], task["connect_tools_all"], n_t,
                      _lib.AG_HEIGHT_MIN, None, raise_by, out=direct)
        if direct is None:
            seq[:, li] = res
    _strict_status(model, device)
    return {"state_seqs": seq, "action_seqs": decoded}


@torch.no_grad()
def dynamics_masked(state_init, state_mask, action, model, device, ppm_optimizer, physics_param=None):
    task = ppm_optimizer.task_config
    n_his, push_length = task["n_his"], task["push_length"]
    state = state_init.to(device, torch.float32)
    state_mask = state_mask.to(device)
    action = action.to(device, torch.float32)
    bsz, n_obj = state.shape[0], state.shape[1]
    decoded, repeat = decode_action(action[:, None], push_length=push_length)
    decoded, repeat = decoded[:, 0], repeat[:, 0]
    n_t = ppm_optimizer.eef_num
    N = n_obj + n_t

    cnt = state_mask.sum(dim=1)
    y = (state[:, :, 1] * state_mask).sum(dim=1) / cnt                            # forward_dynamics.py:235
    eef, dlt, raise_by = _place_tool(task, decoded, action[:, 2], y, device)
    state0 = torch.empty((bsz, n_his, N, 3), device=device)
    state0[:, :, :n_obj] = state[:, None]
    state0[:, :, n_obj:] = eef[:, None]
    delta = torch.zeros((bsz, N, 3), device=device)
    delta[:, n_obj:] = dlt
    attrs = torch.zeros((bsz, N, 2), device=device)
    attrs[:, :n_obj, 0] = state_mask.float()
    attrs[:, n_obj:, 1] = 1.0
    p_instance = torch.zeros((bsz, n_obj, task["max_n"]), device=device)           # first `count` slots, :292-300
    p_instance[:, :, 0] = (torch.arange(n_obj, device=device)[None] < cnt[:, None]).float()
    mask = torch.ones((bsz, N), dtype=torch.bool, device=device)
    mask[:, :n_obj] = state_mask.bool()
    tool_mask = torch.zeros((bsz, N), dtype=torch.bool, device=device)
    tool_mask[:, n_obj:] = True
    phys = _physics(ppm_optimizer, physics_param, bsz, device)
    thr = threshold_sq(ppm_optimizer.adj_thresh, bsz, torch.device(device), _lib.AG_VARIANT_BATCH)
    n_steps = int(repeat.max().item())
    model.take_status(device)
    seq = rollout(model, state0, delta, attrs, p_instance, phys, mask, tool_mask, thr, repeat, n_steps, task["topk"],
                  task["connect_tools_all"], n_t, _lib.AG_HEIGHT_MASKED_MEAN, state_mask.bool(), raise_by)
    _strict_status(model, device)
    return {"state_seqs": seq, "action_seqs": decoded}
