"""MPPI planning iteration on the engine (BASELINE configs[4]).

`running_cost` is a drop-in for src/planning/plan.py:27-59; `MPPIPlanner.trajectory_optimization_mppi` follows
src/planning/real_world/planner.py:234-277 (sample -> rollout -> evaluate -> softmax update -> keep the best sample),
with the rollout sharded over the ranks of the process group when one is initialised.  Everything stays on the GPU:
the only host read per iteration is the `error.max()` normaliser the reference also reads.
"""
import functools

import torch

from . import losses
from .dist import dynamics_sharded, replicate
from .forward_dynamics import dynamics
from .plan_utils import clip_actions, optimize_action_mppi, sample_action_seq


def running_cost(state, action, state_cur, error_func, penalty_func, bbox, **kwargs):
    """state (bsz, L, n, 3), action (bsz, L, 4), state_cur (n, 3), bbox [[xmin,xmax],[zmin,zmax]] -> {"reward_seqs": (bsz,)}."""
    bsz, L = state.shape[0], state.shape[1]
    error = error_func(state.reshape(bsz * L, state.shape[2], state.shape[3])).reshape(bsz, L)
    error_weight = 2.0 / (error.max().item() + 1e-6)
    collision = penalty_func(state, action, state_cur)
    lo, hi = state.min(dim=2).values, state.max(dim=2).values                     # (bsz, L, 3)
    bbox = torch.as_tensor(bbox, dtype=state.dtype, device=state.device)
    margins = torch.stack([lo[..., 0] - bbox[0, 0], bbox[0, 1] - hi[..., 0], lo[..., 2] - bbox[1, 0], bbox[1, 1] - hi[..., 2]], dim=-1)
    box = torch.exp(-margins.clamp_min(0) * 100.0).max(dim=-1).values             # 1 when a particle sits on the workspace edge
    reward = -error_weight * error[:, -1] - 5.0 * collision.mean(dim=1) - 5.0 * box.mean(dim=1)
    return {"reward_seqs": reward}


class MPPIPlanner:
    """Minimal planner with the reference Planner's MPPI branch (planner.py:38-326 keeps many unrelated modes)."""

    def __init__(self, model, device, ppm_optimizer, error_func, penalty_func, bbox, action_lower_lim, action_upper_lim,
                 n_sample, n_look_ahead=1, n_update_iter=1, reward_weight=500.0, noise_level=1.0, rollout_best=True):
        task = ppm_optimizer.task_config
        self.device = device
        self.lo = torch.as_tensor(action_lower_lim, dtype=torch.float32, device=device)
        self.hi = torch.as_tensor(action_upper_lim, dtype=torch.float32, device=device)
        self.n_sample, self.n_look_ahead, self.n_update_iter = n_sample, n_look_ahead, n_update_iter
        self.reward_weight, self.noise_level, self.push_length = reward_weight, noise_level, task["push_length"]
        self.rollout_best = rollout_best
        self.model_rollout = lambda state, acts: dynamics_sharded(dynamics, state, acts, model, device, ppm_optimizer)
        self.evaluate_traj = functools.partial(running_cost, error_func=error_func, penalty_func=penalty_func, bbox=bbox)

    def sample(self, act_seq, iter_index, device=None):
        """Sampled action sequences, REPLICATED across the process group: every rank scores the gathered rollouts of all
        samples against this tensor, so rank 0's draw is broadcast (ranks need not share an RNG state)."""
        dev = self.device if device is None else device
        acts = sample_action_seq(act_seq.to(dev), self.lo.to(dev), self.hi.to(dev), self.n_sample, dev, iter_index=iter_index,
                                 noise_level=self.noise_level, push_length=self.push_length).to(self.device)
        return replicate(acts)

    @torch.no_grad()
    def step(self, state_cur, act_seqs):
        """One MPPI update from GIVEN samples: rollout, rewards, softmax-weighted new sequence."""
        out = self.model_rollout(state_cur, act_seqs)
        reward = self.evaluate_traj(out["state_seqs"], act_seqs, state_cur=state_cur)["reward_seqs"]
        new_seq = optimize_action_mppi(act_seqs, reward, reward_weight=self.reward_weight, action_lower_lim=self.lo,
                                       action_upper_lim=self.hi, push_length=self.push_length)
        return new_seq, reward, out

    @torch.no_grad()
    def trajectory_optimization_mppi(self, state_cur, act_seq, sample_device=None):
        best_seq, best_reward = None, None
        for i in range(self.n_update_iter):
            act_seqs = self.sample(act_seq, i, sample_device)
            act_seq, reward, _ = self.step(state_cur, act_seqs)
            k = torch.argmax(reward)
            if best_reward is None or reward[k] > best_reward:
                best_seq, best_reward = act_seqs[k], reward[k]
        res = {"act_seq": best_seq, "best_reward": best_reward}
        if self.rollout_best:
            out = self.model_rollout(state_cur, best_seq[None])
            res["best_model_output"] = out
            res["best_eval_output"] = self.evaluate_traj(out["state_seqs"], best_seq[None], state_cur=state_cur)
        return res
