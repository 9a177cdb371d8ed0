"""Action codec used by the rollout drivers — drop-in for src/planning/plan_utils.py:11-20."""
import torch


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
