"""ctypes binding + numpy rollout drivers for the CPU oracle (oracle/ag_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from adaptigraph_amd/.  Parity status: pinned against tests/golden/
(generated from the imported reference by tools/gen_golden.py).

The dense/graph arithmetic lives in C; the rollout drivers below restate the reference's
Python loops (src/planning/forward_dynamics.py, src/planning/plan_utils.py) with numpy in fp32.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATE_DICT_KEYS = (
    [f"particle_encoder.model.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [f"relation_encoder.model.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")]
    + [f"particle_propagator.linear.{p}" for p in ("weight", "bias")]
    + [f"relation_propagator.linear.{p}" for p in ("weight", "bias")]
    + [f"non_rigid_predictor.linear_{i}.{p}" for i in (0, 1, 2) for p in ("weight", "bias")]
)


class _Config(ctypes.Structure):
    _fields_ = [("n_his", ctypes.c_int), ("attr_dim", ctypes.c_int), ("phys_dim", ctypes.c_int),
                ("action_dim", ctypes.c_int), ("nf", ctypes.c_int), ("pstep", ctypes.c_int),
                ("n_instance", ctypes.c_int), ("motion_clamp", ctypes.c_float)]


def build(force=False):
    so = os.path.join(_HERE, "libag_oracle.so")
    src = os.path.join(_HERE, "ag_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libag_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.ago_forward.restype = ctypes.c_int
        _LIB.ago_build_edges.restype = ctypes.c_int
    return _LIB


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build_edges(pos, radius, mask, tool_mask, topk, connect_tools_all, variant, e_cap=None):
    """-> (n_rel (B,), recv (B,e_cap), send (B,e_cap)); variant 'single' | 'batch' (graph.py:38-156)."""
    pos = _f32(pos)
    B, N, _ = pos.shape
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    tool_mask = np.ascontiguousarray(tool_mask, dtype=np.uint8)
    radius = np.ascontiguousarray(np.broadcast_to(np.asarray(radius, np.float64), (B,)))
    if e_cap is None:
        e_cap = N * (min(N, topk) + int(tool_mask.sum(1).max()))
    recv = np.full((B, e_cap), -1, np.int32)
    send = np.full((B, e_cap), -1, np.int32)
    n_rel = np.zeros(B, np.int32)
    rc = lib().ago_build_edges(_p(pos, ctypes.c_float), _p(mask, ctypes.c_uint8), _p(tool_mask, ctypes.c_uint8),
                               _p(radius, ctypes.c_double), int(topk), int(bool(connect_tools_all)),
                               0 if variant == "single" else 1, B, N, e_cap, _p(recv, ctypes.c_int32),
                               _p(send, ctypes.c_int32), _p(n_rel, ctypes.c_int32))
    if rc != 0:
        raise RuntimeError(f"ago_build_edges failed: {rc}")
    return n_rel, recv, send


def forward(weights, state, attrs, action, p_instance, phys, n_rel, recv, send, pstep=3, motion_clamp=100.0):
    """weights: dict keyed like the reference state_dict.  -> (pred_pos, pred_motion) (B,n_p,3)."""
    state, attrs, action, p_instance, phys = map(_f32, (state, attrs, action, p_instance, phys))
    B, H, N, _ = state.shape
    n_p, n_inst = p_instance.shape[1], p_instance.shape[2]
    recv = np.ascontiguousarray(recv, np.int32)
    send = np.ascontiguousarray(send, np.int32)
    n_rel = np.ascontiguousarray(n_rel, np.int32)
    ws = [_f32(weights[k]) for k in STATE_DICT_KEYS]
    nf = ws[0].shape[0]
    cfg = _Config(H, attrs.shape[2], phys.shape[1], action.shape[2], nf, pstep, n_inst, motion_clamp)
    assert ws[0].shape[1] == attrs.shape[2] + phys.shape[1] + action.shape[2]
    assert ws[6].shape[1] == 2 * attrs.shape[2] + 1 + 3 * H
    arr = (ctypes.POINTER(ctypes.c_float) * len(ws))(*[_p(w, ctypes.c_float) for w in ws])
    pred_pos = np.zeros((B, n_p, 3), np.float32)
    pred_motion = np.zeros((B, n_p, 3), np.float32)
    rc = lib().ago_forward(ctypes.byref(cfg), arr, _p(state, ctypes.c_float), _p(attrs, ctypes.c_float),
                           _p(action, ctypes.c_float), _p(p_instance, ctypes.c_float), _p(phys, ctypes.c_float),
                           _p(recv, ctypes.c_int32), _p(send, ctypes.c_int32), _p(n_rel, ctypes.c_int32),
                           recv.shape[1], B, N, n_p, _p(pred_pos, ctypes.c_float), _p(pred_motion, ctypes.c_float))
    if rc != 0:
        raise RuntimeError(f"ago_forward failed: {rc}")
    return pred_pos, pred_motion


def decode_action(action, push_length=0.10):
    """plan_utils.py:11-20.  [x, z, theta, len] -> [x_s, z_s, x_e, z_e], repeat = int32(len) (truncation)."""
    action = _f32(action)
    pl = np.float32(push_length)
    x_s, z_s, th = action[..., 0], action[..., 1], action[..., 2]
    repeat = action[..., 3].astype(np.int32)
    x_e = x_s - pl * np.cos(th)
    z_e = z_s - pl * np.sin(th)
    return np.stack([x_s, z_s, x_e, z_e], -1).astype(np.float32), repeat


def _place_tool(task, decoded, theta, y):
    """Tool key-points + per-step delta from one decoded action (forward_dynamics.py:42-81 / :237-276)."""
    bsz = decoded.shape[0]
    pp = task["pusher_points"]
    ratio = task["sim_real_ratio"]
    n_t = len(pp)
    eef = np.zeros((bsz, n_t, 3), np.float32)
    delta = np.zeros((bsz, n_t, 3), np.float32)
    delta[:, :, 0] = (decoded[:, 2] - decoded[:, 0])[:, None]
    delta[:, :, 2] = (decoded[:, 3] - decoded[:, 1])[:, None]
    if n_t == 1:
        eef[:, 0, 0] = decoded[:, 0]
        eef[:, 0, 2] = decoded[:, 1]
    elif n_t == 5:
        s, c = np.sin(theta), np.cos(theta)
        for i in range(5):
            off = np.float32(float(pp[i][1]) * ratio) if i else np.float32(0.0)
            eef[:, i, 0] = decoded[:, 0] + off * s if i else decoded[:, 0]
            eef[:, i, 2] = decoded[:, 1] - off * c if i else decoded[:, 1]
    else:
        raise NotImplementedError("pusher not implemented")   # forward_dynamics.py:78
    eef[:, :, 1] = y[:, None]
    if task["gripper_enable"]:
        eef[:, :, 1] += np.float32(0.01 * ratio)
    return eef, delta


def _rollout_core(weights, task, states, delta, attrs, p_instance, phys, mask, tool_mask, radius, repeat, n_obj,
                  height_fn, pstep, trace=None):
    """Inner loop shared by dynamics / dynamics_masked (forward_dynamics.py:156-197 / :351-393).
    `trace` (a list) receives, per model step, the inputs of that step: the state history and the edge lists built on
    its newest frame (tests use it to teacher-force the engine along the reference trajectory)."""
    bsz = states.shape[0]
    out = np.zeros((bsz, n_obj, 3), np.float32)
    topk, connect = task["topk"], task["connect_tools_all"]
    n_rel, recv, send = build_edges(states[:, -1], radius, mask, tool_mask, topk, connect, "batch")
    for ai in range(1, 1 + int(repeat.max())):
        if trace is not None:
            trace.append(dict(states=states.copy(), delta=delta, attrs=attrs, p_instance=p_instance, phys=phys, mask=mask,
                              tool_mask=tool_mask, radius=radius, n_rel=n_rel.copy(), recv=recv.copy(), send=send.copy()))
        pred, _ = forward(weights, states, attrs, delta, p_instance, phys, n_rel, recv, send, pstep=pstep)
        sel = repeat == ai
        out[sel] = pred[sel]
        eef = states[:, -1, n_obj:] + delta[:, n_obj:]
        eef[:, :, 1] = height_fn(pred)[:, None]
        if task["gripper_enable"]:
            eef[:, :, 1] += np.float32(0.01 * task["sim_real_ratio"])
        cur = np.concatenate([pred, eef], 1)
        n_rel, recv, send = build_edges(cur, radius, mask, tool_mask, topk, connect, "batch")
        states = np.concatenate([states[:, 1:], cur[:, None]], 1)
    return out


def dynamics(weights, task, state, action, radius=None, phys_value=0.5, pstep=3, trace=None):
    """forward_dynamics.py:11-205.  state (n_obj,3), action (bsz,L,4) -> state_seqs (bsz,L,n_obj,3), action_seqs."""
    state, action = _f32(state), _f32(action)
    bsz, L = action.shape[:2]
    n_his, n_t = task["n_his"], task["eef_num"]
    decoded, repeat = decode_action(action, task["push_length"])
    n_obj = state.shape[0]
    N = n_obj + n_t
    radius = task["adj_thresh"] if radius is None else radius
    obj = np.broadcast_to(state, (bsz, n_his, n_obj, 3)).copy()
    seq = np.zeros((bsz, L, n_obj, 3), np.float32)
    attrs = np.zeros((bsz, N, 2), np.float32)
    attrs[:, :n_obj, 0] = 1
    attrs[:, n_obj:, 1] = 1
    p_instance = np.zeros((bsz, n_obj, task["max_n"]), np.float32)
    p_instance[:, :, 0] = 1
    mask = np.ones((bsz, N), bool)
    tool_mask = np.zeros((bsz, N), bool)
    tool_mask[:, n_obj:] = True
    phys = np.full((bsz, 1), phys_value, np.float32)
    for li in range(L):
        if li > 0:
            obj = np.repeat(seq[:, li - 1:li], n_his, 1)
        y = obj[:, -1, :, 1].min(1)
        eef, delta_t = _place_tool(task, decoded[:, li], action[:, li, 2], y)
        states = np.zeros((bsz, n_his, N, 3), np.float32)
        states[:, :, :n_obj] = obj
        states[:, :, n_obj:] = eef[:, None]
        delta = np.zeros((bsz, N, 3), np.float32)
        delta[:, n_obj:] = delta_t
        seq[:, li] = _rollout_core(weights, task, states, delta, attrs, p_instance, phys, mask, tool_mask, radius,
                                   repeat[:, li], n_obj, lambda pred: pred[:, :, 1].min(1), pstep, trace)
    return seq, decoded


def dynamics_masked(weights, task, state_init, state_mask, action, radius=None, phys_value=0.5, pstep=3, trace=None):
    """forward_dynamics.py:208-399.  state_init (bsz,n,3), state_mask (bsz,n) bool, action (bsz,4)."""
    state_init, action = _f32(state_init), _f32(action)
    state_mask = np.asarray(state_mask, bool)
    bsz, n_obj = state_init.shape[:2]
    n_his, n_t = task["n_his"], task["eef_num"]
    decoded, repeat = decode_action(action[:, None], task["push_length"])
    decoded, repeat = decoded[:, 0], repeat[:, 0]
    N = n_obj + n_t
    radius = task["adj_thresh"] if radius is None else radius
    cnt = state_mask.sum(1).astype(np.float32)

    def height(p):   # masked mean y, forward_dynamics.py:235,359
        return (p[:, :, 1] * state_mask).sum(1, dtype=np.float32) / cnt

    eef, delta_t = _place_tool(task, decoded, action[:, 2], height(state_init))
    states = np.zeros((bsz, n_his, N, 3), np.float32)
    states[:, :, :n_obj] = state_init[:, None]
    states[:, :, n_obj:] = eef[:, None]
    delta = np.zeros((bsz, N, 3), np.float32)
    delta[:, n_obj:] = delta_t
    attrs = np.zeros((bsz, N, 2), np.float32)
    attrs[:, :n_obj, 0] = state_mask
    attrs[:, n_obj:, 1] = 1
    p_instance = np.zeros((bsz, n_obj, task["max_n"]), np.float32)
    for b in range(bsz):
        p_instance[b, :int(state_mask[b].sum()), 0] = 1
    mask = np.ones((bsz, N), bool)
    mask[:, :n_obj] = state_mask
    tool_mask = np.zeros((bsz, N), bool)
    tool_mask[:, n_obj:] = True
    phys = np.full((bsz, 1), phys_value, np.float32)
    seq = _rollout_core(weights, task, states, delta, attrs, p_instance, phys, mask, tool_mask, radius, repeat, n_obj,
                        height, pstep, trace)
    return seq, decoded


# ---------------------------------------------------------------------------------------------------------------
# MPPI glue — numpy restatements of src/planning/losses.py, plan.py:27-59 and
# plan_utils.py:31-39,80-101, pinned to tests/golden/mppi_*.npz.
# ---------------------------------------------------------------------------------------------------------------
def chamfer(x, y):
    """losses.py:4-10: x (B,N,3), y (B|1,M,3) -> (B,)."""
    x, y = _f32(x), _f32(y)
    dis = np.sqrt(((x[:, None, :, :] - y[:, :, None, :]) ** 2).sum(-1, dtype=np.float32))       # (B, M, N)
    return dis.min(2).mean(1, dtype=np.float32) + dis.min(1).mean(1, dtype=np.float32)


def box_loss(state, target):
    """losses.py:26-35."""
    x, z = state[:, :, 0], state[:, :, 2]
    dx = np.maximum(target[0, 0] - x, 0) + np.maximum(x - target[0, 1], 0)
    dz = np.maximum(target[1, 0] - z, 0) + np.maximum(z - target[1, 1], 0)
    return np.sqrt(dx ** 2 + dz ** 2).mean(1, dtype=np.float32)


def _pre_push_states(state_pred, state_init):
    first = np.broadcast_to(state_init[:, [0, 2]][None, None], (state_pred.shape[0], 1) + state_init[:, [0, 2]].shape)
    return np.concatenate([first, state_pred[:, :-1][..., [0, 2]]], 1)


def rope_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:37-48."""
    d = np.linalg.norm(action[:, :, None, 0:2] - _pre_push_states(state_pred, state_init), axis=-1).min(-1)
    return np.exp(-np.maximum(d - np.float32(0.02 * sim_real_ratio), 0) * np.float32(100.0)).astype(np.float32)


def cloth_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:50-64."""
    d = np.linalg.norm(action[:, :, None, 0:2] - state_init[None, None][..., [0, 2]], axis=-1)
    dmin = np.maximum(d.min(-1) - np.float32(0.005 * sim_real_ratio), 0)
    dmax = np.minimum(d.max(-1), np.float32(0.4 * sim_real_ratio))
    dmax = dmax / dmax.max()
    return (1.0 - np.exp(-dmin * 100.0) - dmax * 0.2).astype(np.float32)


def granular_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """losses.py:66-92."""
    x0, z0, th = action[:, :, 0], action[:, :, 1], action[:, :, 2]
    rad = np.float32(0.05 * sim_real_ratio)
    dx, dz = rad * np.sin(th), -rad * np.cos(th)
    offs = np.array([-1, -0.75, -0.5, -0.25, 0, 0.25, 0.5, 0.75, 1], np.float32)
    pts = np.stack([x0[..., None] + offs * dx[..., None], z0[..., None] + offs * dz[..., None]], -1)
    s2d = _pre_push_states(state_pred, state_init)
    d = np.linalg.norm(pts[:, :, :, None] - s2d[:, :, None], axis=-1).min(-1).min(-1)
    return np.exp(-np.maximum(d - np.float32(0.02 * sim_real_ratio), 0) * np.float32(100.0)).astype(np.float32)


def running_cost(state, action, state_cur, error_func, penalty_func, bbox):
    """plan.py:27-59 -> reward (bsz,)."""
    bsz, L = state.shape[:2]
    error = error_func(state.reshape(bsz * L, state.shape[2], 3)).reshape(bsz, L)
    w = 2.0 / (float(error.max()) + 1e-6)
    pen = penalty_func(state, action, state_cur)
    lo, hi = state.min(2), state.max(2)
    m = np.stack([lo[..., 0] - bbox[0, 0], bbox[0, 1] - hi[..., 0], lo[..., 2] - bbox[1, 0], bbox[1, 1] - hi[..., 2]], -1)
    box = np.exp(-np.maximum(m, 0) * 100.0).max(-1)
    return (-w * error[:, -1] - 5.0 * pen.mean(1) - 5.0 * box.mean(1)).astype(np.float32)


def clip_actions(action, lo, hi):
    """plan_utils.py:31-39."""
    out = np.array(action, np.float32)
    out[..., 2] = ((out[..., 2] + np.float32(np.pi)) % np.float32(2 * np.pi)) - np.float32(np.pi)
    return np.clip(out, lo, hi)


def optimize_action_mppi(act_seqs, reward, reward_weight, lo, hi, push_length):
    """plan_utils.py:80-101."""
    z = reward.astype(np.float32) * np.float32(reward_weight)
    w = np.exp(z - z.max())
    w = (w / w.sum())[:, None]
    xs, zs, th, ln = (act_seqs[..., k] for k in range(4))
    pl = np.float32(push_length)
    xe, ze = xs - ln * pl * np.cos(th), zs - ln * pl * np.sin(th)
    x, zz, x_e, z_e = ((w * v).sum(0) for v in (xs, zs, xe, ze))
    theta = np.arctan2(zz - z_e, x - x_e)
    length = np.hypot(x_e - x, z_e - zz) / pl
    return clip_actions(np.stack([x, zz, theta, length], -1), lo, hi)


# ---------------------------------------------------------------------------------------------------------------
# Sys-id objective: losses.py:12-24 + physics_param_optimizer.py:178-226, pinned to
# tests/golden/sysid_*.npz.
# ---------------------------------------------------------------------------------------------------------------
def mean_chamfer(state_pred, state_real, pred_mask, real_mask):
    return np.array([float(chamfer(state_pred[i][pred_mask[i].astype(bool)][None], state_real[i][real_mask[i].astype(bool)][None])[0])
                     for i in range(state_pred.shape[0])], np.float64)


def dynamics_error(weights, task, phys_value, state_init_list, state_real_list, actions):
    """-> (mean error, per-interaction errors, predicted padded states)."""
    mx, n = task["max_nobj"], len(actions)
    pad = lambda L: np.stack([np.pad(_f32(x), ((0, mx - len(x)), (0, 0))) for x in L])
    msk = lambda L: np.stack([np.arange(mx) < len(x) for x in L])
    seq, _ = dynamics_masked(weights, task, pad(state_init_list), msk(state_init_list), np.stack(actions), phys_value=phys_value)
    per = mean_chamfer(seq, pad(state_real_list), msk(state_init_list), msk(state_real_list))
    return float(per.mean()), per, seq


# ---------------------------------------------------------------------------------------------------------------
# Open-loop evaluation rollout: the per-graph loop of src/dynamics/rollout/rollout.py:66-141 for a
# given start graph and frame schedule, pinned to tests/golden/evalrollout_rope.npz.
# ---------------------------------------------------------------------------------------------------------------
def eval_rollout(weights, graph, fps_idx, schedule, eef_pos, obj_pos, adj_thresh, topk, connect_tool_all, phys):
    state = _f32(graph["state"])[None]
    action = _f32(graph["action"])[None]
    attrs, p_inst = _f32(graph["attrs"])[None], _f32(graph["p_instance"])[None]
    smask, emask, omask = graph["state_mask"][None], graph["eef_mask"][None], np.asarray(graph["obj_mask"], bool)
    max_nobj = p_inst.shape[1]
    errors = []
    for t, (s, e) in enumerate(schedule):
        n_rel, recv, send = build_edges(state[:, -1], adj_thresh, smask, emask, topk, connect_tool_all, "single")
        pred, _ = forward(weights, state, attrs, action, p_inst, _f32(phys)[None], n_rel, recv, send)
        gt = np.zeros((max_nobj, 3), np.float32)
        gt[:len(fps_idx)] = obj_pos[e][fps_idx]
        errors.append(float(np.linalg.norm(pred[0][omask] - gt[omask], axis=-1).mean()))
        if t + 1 < len(schedule):
            s2, e2 = schedule[t + 1]
            cur = np.concatenate([pred[0], eef_pos[s2]], 0)
            state = np.concatenate([state[:, 1:], cur[None, None]], 1)
            action = np.zeros_like(action)
            action[0, max_nobj:] = eef_pos[e2] - eef_pos[s2]
    return errors


# ---------------------------------------------------------------------------------------------------------------
# Training objective: the n_future unroll of src/dynamics/train/train.py:84-108, forward only,
# pinned to tests/golden/train_rope.npz (the gradients there come from the reference's autograd).
# ---------------------------------------------------------------------------------------------------------------
def unrolled_loss(weights, batch, n_rel, recv, send, n_future=3):
    state, action = _f32(batch["state"]).copy(), _f32(batch["action"]).copy()
    n_p = batch["p_instance"].shape[1]
    loss, preds = 0.0, []
    for fi in range(n_future):
        pred, _ = forward(weights, state, batch["attrs"], action, batch["p_instance"], batch["rope_physics_param"], n_rel, recv, send)
        preds.append(pred)
        loss += float(np.mean((pred - batch["state_future"][:, fi]) ** 2, dtype=np.float64))
        if fi < n_future - 1:
            nxt = _f32(batch["eef_future"][:, fi]).copy()
            nxt[:, :n_p] = pred
            state = np.concatenate([state[:, 1:], nxt[:, None]], 1)
            action = _f32(batch["action_future"][:, fi])
    return loss, np.stack(preds)
