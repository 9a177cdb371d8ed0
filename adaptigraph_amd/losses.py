"""Trajectory cost terms of the planner — drop-in for src/planning/losses.py.

`chamfer` runs the HIP kernel (`ag_chamfer`: no (B,M,N,3) temporaries); the penalties are a few element-wise ops
over (bsz, n_look_forward, n_obj) and stay as device tensor ops.
"""
import ctypes

import torch

from . import _lib
from .graph import _require_gpu, _stream_ptr


def chamfer(x, y):
    """x (B,N,3), y (B or 1,M,3) -> (B,)  mean_m min_n ||x-y|| + mean_n min_m ||x-y||   (losses.py:4-10)."""
    _require_gpu(x, "x")
    assert x.dim() == 3 and y.dim() == 3 and x.shape[2] == 3 and y.shape[2] == 3
    assert y.shape[0] in (1, x.shape[0])
    x = x.contiguous().float()
    y = y.to(x.device).contiguous().float()
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ag_chamfer(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], y.shape[1],
                                   1 if (y.shape[0] == x.shape[0] and x.shape[0] > 1) else 0, out.data_ptr(), _stream_ptr(x.device))
    _lib.check(rc, "ag_chamfer")
    return out


def mean_chamfer_device(state_pred, state_real, state_pred_mask, state_real_mask):
    """Per-sample chamfer over the masked-in points of two padded clouds, one launch for the whole batch -> (bsz,) tensor."""
    _require_gpu(state_pred, "state_pred")
    dev = state_pred.device
    x = state_pred.contiguous().float()
    y = state_real.to(dev).contiguous().float()
    xm = state_pred_mask.to(dev).ne(0).to(torch.uint8).contiguous()
    ym = state_real_mask.to(dev).ne(0).to(torch.uint8).contiguous()
    assert x.dim() == 3 and y.dim() == 3 and y.shape[0] == x.shape[0] and xm.shape == x.shape[:2] and ym.shape == y.shape[:2]
    out = torch.empty(x.shape[0], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().ag_chamfer_masked(x.data_ptr(), xm.data_ptr(), y.data_ptr(), ym.data_ptr(), x.shape[0], x.shape[1],
                                          y.shape[1], 1, out.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "ag_chamfer_masked")
    return out


def mean_chamfer(state_pred, state_real, state_pred_mask, state_real_mask):
    """losses.py:12-24: numpy (bsz,) of chamfer(state_pred[i][mask_i], state_real[i][mask_i]); the reference loops over
    the batch with one `.item()` sync per sample, here it is one kernel and one copy."""
    return mean_chamfer_device(state_pred, state_real, state_pred_mask, state_real_mask).double().cpu().numpy()


def box_loss(state, target):
    """state (B,N,3), target [[xmin,xmax],[zmin,zmax]] -> (B,) mean distance to the box in the x-z plane (losses.py:26-35)."""
    x, z = state[:, :, 0], state[:, :, 2]
    dx = (target[0, 0] - x).clamp_min(0) + (x - target[0, 1]).clamp_min(0)
    dz = (target[1, 0] - z).clamp_min(0) + (z - target[1, 1]).clamp_min(0)
    return ((dx ** 2 + dz ** 2) ** 0.5).mean(dim=1)


def _states_before_push(state_pred, state_init):
    """x-z particle positions at the START of every look-ahead push: [state_init, state_pred[:, :-1]] (losses.py:42-43)."""
    bsz = state_pred.shape[0]
    first = state_init[:, [0, 2]][None, None].expand(bsz, 1, -1, -1)
    return torch.cat([first, state_pred[:, :-1, :, [0, 2]]], dim=1)


def rope_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """exp(-100 max(d - 0.02 r, 0)) with d the pusher-start to nearest-particle distance (losses.py:37-48)."""
    pts = action[:, :, 0:2]                                            # (bsz, L, 2) = (x_start, z_start)
    d = (pts[:, :, None] - _states_before_push(state_pred, state_init)).norm(dim=-1).min(dim=-1).values
    return torch.exp(-(d - 0.02 * sim_real_ratio).clamp_min(0) * 100.0)


def cloth_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """Grasp point must touch the cloth (min distance) and prefers far-from-edge picks (max distance) (losses.py:50-64)."""
    pts = action[:, :, 0:2]
    d = (pts[:, :, None] - state_init[:, [0, 2]][None, None]).norm(dim=-1)          # (bsz, L, n)
    dmin = (d.min(dim=-1).values - 0.005 * sim_real_ratio).clamp_min(0)
    dmax = d.max(dim=-1).values.clamp_max(0.4 * sim_real_ratio)
    dmax = dmax / dmax.max()
    return 1.0 - torch.exp(-dmin * 100.0) - dmax * 0.2


def granular_penalty(state_pred, action, state_init, sim_real_ratio=10.0):
    """Nine points along the flat pusher (half-width 0.05 r) must not start inside the pile (losses.py:66-92)."""
    x0, z0, theta = action[:, :, 0], action[:, :, 1], action[:, :, 2]
    rad = 0.05 * sim_real_ratio
    dx, dz = rad * torch.sin(theta), -rad * torch.cos(theta)
    offs = torch.tensor([-1.0, -0.75, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75, 1.0], device=action.device)
    pts = torch.stack([x0[..., None] + offs * dx[..., None], z0[..., None] + offs * dz[..., None]], dim=-1)   # (bsz, L, 9, 2)
    s2d = _states_before_push(state_pred, state_init)                                                         # (bsz, L, n, 2)
    d = (pts[:, :, :, None] - s2d[:, :, None]).norm(dim=-1).min(dim=-1).values.min(dim=-1).values
    return torch.exp(-(d - 0.02 * sim_real_ratio).clamp_min(0) * 100.0)
