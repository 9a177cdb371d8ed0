"""Action codec and MPPI sampling/update — drop-in for src/planning/plan_utils.py:11-101."""
import math

import torch
import torch.nn.functional as F


def decode_action(action, push_length=0.10):
    """(…, 4) [x, z, theta, length] -> ((…, 4) [x_s, z_s, x_e, z_e], (…) int32 repeat).

    The push end point is one `push_length` step along -(cos, sin)(theta); `length` is truncated toward zero
    to the number of model steps the push is repeated for (plan_utils.py:15-16)."""
    start = action[..., 0:2]
    theta = action[..., 2]
    repeat = action[..., 3].detach().to(torch.int32)
    direction = torch.stack([torch.cos(theta), torch.sin(theta)], dim=-1)
    end = start - push_length * direction
    return torch.cat([start, end], dim=-1), repeat


def angle_normalize(x):
    return ((x + math.pi) % (2 * math.pi)) - math.pi


def clip_actions(action, action_lower_lim, action_upper_lim):
    """Wrap theta into [-pi, pi) and clamp every field to its limits (plan_utils.py:31-39)."""
    out = action.clone()
    out[..., 2] = angle_normalize(action[..., 2])
    return torch.maximum(torch.minimum(out, action_upper_lim), action_lower_lim)


def _push_ends(act_seqs, push_length):
    xs, zs, th, ln = act_seqs.unbind(-1)
    return xs, zs, xs - ln * push_length * torch.cos(th), zs - ln * push_length * torch.sin(th)


def _from_ends(xs, zs, xe, ze, push_length):
    theta = torch.atan2(zs - ze, xs - xe)
    length = torch.stack([xe - xs, ze - zs], dim=-1).norm(dim=-1) / push_length
    return torch.stack([xs, zs, theta, length], dim=-1)


def sample_action_seq(act_seq, action_lower_lim, action_upper_lim, n_sample, device, iter_index=0, noise_level=0.3,
                      push_length=0.10):
    """(L,4) -> (n_sample, L, 4).  iter 0: uniform in the limits; later: Gaussian noise on both push end points, sigma
    growing 10x per look-ahead index, sample 0 kept as the unperturbed sequence (plan_utils.py:42-77).  Consumes the
    torch RNG exactly as the reference does (one rand / one normal(n_sample,4) per look-ahead step)."""
    L = act_seq.shape[0]
    if iter_index == 0:
        return torch.rand((n_sample, L, act_seq.shape[1]), device=device) * (action_upper_lim - action_lower_lim) + action_lower_lim
    assert act_seq.shape[-1] == 4
    out = act_seq.to(device)[None].repeat(n_sample, 1, 1)
    xs, zs, xe, ze = _push_ends(out, push_length)
    for i in range(L):
        res = 0.1 * (10 ** i) * torch.normal(0, noise_level, (n_sample, 4), device=device)
        cand = _from_ends(xs[:, i] + res[:, 0], zs[:, i] + res[:, 1], xe[:, i] + res[:, 2], ze[:, i] + res[:, 3], push_length)
        out[1:, i] = clip_actions(cand, action_lower_lim, action_upper_lim)[1:]
    return out


def optimize_action_mppi(act_seqs, reward_seqs, reward_weight=100.0, action_lower_lim=None, action_upper_lim=None,
                         push_length=0.10):
    """Softmax(reward * weight)-weighted mean of the push START and END points, re-encoded as (x, z, theta, length)
    and clipped (plan_utils.py:80-101)."""
    assert act_seqs.shape[-1] == 4
    w = F.softmax(reward_seqs * reward_weight, dim=0)[:, None]
    xs, zs, xe, ze = _push_ends(act_seqs, push_length)
    mean = [torch.sum(w * v, dim=0) for v in (xs, zs, xe, ze)]
    return clip_actions(_from_ends(*mean, push_length), action_lower_lim, action_upper_lim)
