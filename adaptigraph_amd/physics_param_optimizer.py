"""Online physics-parameter identification on the engine — SURVEY.md §8f row n2.

Mirrors src/planning/physics_param_optimizer.py: `dynamics_error` (:178-226) is the objective — roll every recorded
interaction forward with a candidate parameter (`dynamics_masked`) and average the masked chamfer distance to the observed
clouds — and `PhysicsParamOnlineOptimizer` keeps the interaction-file / `ppo_<i>.npz` contract (:33-72).

The reference minimises the objective with skopt's GP-BO (1 parameter) or CMA-ES (>1), ≈50 sequential evaluations of a
batch of a handful of graphs each.  Neither package exists here, and a sequential black-box search is the wrong shape for
this engine anyway: graphs in a batch are independent, so ALL candidates of a search generation are evaluated in ONE rollout
of `candidates x interactions` graphs (`dynamics_error_batch`).  `optimize` therefore does a dense batched sweep of
[-0.2, 1.2] followed by batched zoom refinements (1-D), or a batched cross-entropy search (>1-D): same objective, same
bounds, same outputs, different (derivative-free) search.
"""
import copy
import glob
import os

import numpy as np
import torch

from .forward_dynamics import dynamics_masked
from .losses import mean_chamfer_device

PARAM_LO, PARAM_HI = -0.2, 1.2      # gp_minimize bounds / CMA bounds of the reference (:99, :151)


def _pad_interactions(ppm_optimizer, state_init_list, state_real_list, actions):
    """Ragged clouds -> (len_act, max_nobj, 3) tensors + masks, actions -> (len_act, 4)   (:190-216)."""
    device = ppm_optimizer.device
    max_nobj = ppm_optimizer.task_config["max_nobj"]
    n = len(actions)
    init = np.zeros((n, max_nobj, 3), np.float32)
    real = np.zeros((n, max_nobj, 3), np.float32)
    init_mask = np.zeros((n, max_nobj), bool)
    real_mask = np.zeros((n, max_nobj), bool)
    for i in range(n):
        a, b = np.asarray(state_init_list[i]), np.asarray(state_real_list[i])
        if a.shape[0] > max_nobj or b.shape[0] > max_nobj:      # np.pad with a negative width raises in the reference
            raise ValueError(f"interaction {i}: more than max_nobj={max_nobj} particles")
        init[i, :a.shape[0]], init_mask[i, :a.shape[0]] = a, True
        real[i, :b.shape[0]], real_mask[i, :b.shape[0]] = b, True
    to = lambda x: torch.from_numpy(x).to(device)
    act = torch.from_numpy(np.stack([np.asarray(a, np.float32) for a in actions], 0)).to(device)
    return to(init), to(init_mask), to(real), to(real_mask), act


@torch.no_grad()
def dynamics_error_batch(candidates, ppm_optimizer, state_init_list, state_real_list, actions):
    """candidates (C, d) -> (C,) mean error of each candidate, all C x len(actions) graphs in one rollout."""
    device = ppm_optimizer.device
    assert len(ppm_optimizer.material_dims) == 1, "only support single material now"
    material = next(iter(ppm_optimizer.material_dims))
    cand = torch.as_tensor(np.asarray(candidates, np.float32), device=device).reshape(len(candidates), -1)
    C, n = cand.shape[0], len(actions)
    init, init_mask, real, real_mask, act = _pad_interactions(ppm_optimizer, state_init_list, state_real_list, actions)
    rep = lambda x: x[None].expand((C,) + tuple(x.shape)).reshape((C * n,) + tuple(x.shape[1:]))
    out = dynamics_masked(rep(init), rep(init_mask), rep(act), ppm_optimizer.model, device, ppm_optimizer,
                          physics_param={material: cand[:, None, :].expand(C, n, -1).reshape(C * n, -1)})
    err = mean_chamfer_device(out["state_seqs"], rep(real), rep(init_mask), rep(real_mask))
    return err.reshape(C, n).double().mean(dim=1).cpu().numpy()


def dynamics_error(physics_param, ppm_optimizer, state_init_list, state_real_list, actions):
    """The reference objective (:178-226): float mean over interactions of the masked chamfer error for ONE parameter."""
    physics_param = copy.deepcopy(physics_param)
    device = ppm_optimizer.device
    if isinstance(physics_param, (list, np.ndarray)):
        assert len(ppm_optimizer.material_dims) == 1, "only support single material now"
        material = next(iter(ppm_optimizer.material_dims))
        physics_param = {material: torch.tensor(physics_param, dtype=torch.float32, device=device)}
    init, init_mask, real, real_mask, act = _pad_interactions(ppm_optimizer, state_init_list, state_real_list, actions)
    physics_param = {k: v.reshape(-1) for k, v in physics_param.items()}       # optimize_cma passes (1, d)
    out = dynamics_masked(init, init_mask, act, ppm_optimizer.model, device, ppm_optimizer, physics_param=physics_param)
    err = mean_chamfer_device(out["state_seqs"].detach(), real, init_mask, real_mask)
    return float(err.double().mean().item())


def optimize(ppm_optimizer, actions, state_init_list, state_pred_list, state_real_list, num_optim_trials=1, iter_idx=0,
             iterations=50, return_res=False, seed=42):
    """Same contract as the reference's `optimize` / `optimize_cma` (:75-175): -> (ppm, error, init_error[, res]).
    `iterations` bounds the number of objective evaluations PER INTERACTION SET exactly as n_calls does there; they are
    spent in batched generations."""
    dims = sum(ppm_optimizer.material_dims.values())
    material = ppm_optimizer.material
    init_param = ppm_optimizer.physics_param[material].detach().float().cpu().numpy().reshape(-1)
    ev = lambda c: dynamics_error_batch(c, ppm_optimizer, state_init_list, state_real_list, actions)
    init_error = float(ev(init_param[None])[0])
    if iterations == 0:
        return init_error
    if iterations < 0:
        iterations = 200
    history_x, history_f = [], []
    if dims == 1:
        per_gen = max(5, iterations // 3)
        lo, hi = PARAM_LO, PARAM_HI
        for _ in range(3):                                   # sweep, then zoom twice around the incumbent
            xs = np.linspace(lo, hi, per_gen, dtype=np.float64)[:, None]
            fs = ev(xs)
            history_x.append(xs); history_f.append(fs)
            k = int(np.argmin(fs))
            step = (hi - lo) / (per_gen - 1)
            lo, hi = max(PARAM_LO, xs[k, 0] - step), min(PARAM_HI, xs[k, 0] + step)
    else:
        rng = np.random.default_rng(seed)
        pop = max(8, iterations // 5)
        mean, std = init_param.astype(np.float64), np.full(dims, 0.2)          # CMA sigma0 of the reference (:147)
        for _ in range(5):
            xs = np.clip(mean + std * rng.standard_normal((pop, dims)), PARAM_LO, PARAM_HI)
            fs = ev(xs)
            history_x.append(xs); history_f.append(fs)
            elite = xs[np.argsort(fs)[: max(2, pop // 4)]]
            mean, std = elite.mean(0), elite.std(0) + 1e-3
    X, F = np.concatenate(history_x), np.concatenate(history_f)
    best = X[int(np.argmin(F))].astype(np.float32)
    error = dynamics_error(best.tolist(), ppm_optimizer, state_init_list, state_real_list, actions)
    res = {"x_iters": X, "func_vals": F, "x": best, "fun": error}
    return (best, error, init_error, res) if return_res else (best, error, init_error)


optimize_cma = optimize      # one batched search covers both of the reference's entry points


class PhysicsParamOnlineOptimizer:
    """Same attributes and file contract as the reference class (:18-72)."""

    def __init__(self, task_config, model, material, device, save_dir):
        self.task_config = task_config
        self.model = model
        self.material = material
        self.device = device
        self.save_dir = save_dir
        self.physics_param = dict()
        self.material_indices = task_config["material_indices"]
        self.material_dims = task_config["material_dims"]
        self.fps_radius = task_config["fps_radius"]
        self.adj_thresh = task_config["adj_thresh"]
        self.eef_num = task_config["eef_num"]
        self.physics_param[self.material] = torch.tensor([0.5], device=device).repeat(self.material_dims[self.material])

    def optimize(self, i, iterations=50):
        files = sorted(glob.glob(os.path.join(self.save_dir, "interaction_*.npz")))
        assert len(files) == i + 1, f"interaction list {len(files)} != {i + 1}"
        act, state_init, state_pred, state_real = [], [], [], []
        for f in files:
            res = np.load(f)
            act.append(res["act"]); state_init.append(res["state_init"])
            state_pred.append(res["state_pred"]); state_real.append(res["state_real"])
        assert iterations > 0
        ppm, error, error_init, _ = optimize(self, act, state_init, state_pred, state_real, iter_idx=i, iterations=iterations,
                                             return_res=True)
        self.physics_param[self.material] = torch.tensor(ppm, dtype=torch.float32, device=self.device).clamp(-0.2, 1.2)
        np.savez(os.path.join(self.save_dir, f"ppo_{i}.npz"), physics_param=np.array(ppm), error=error, error_init=error_init)
        return ppm, error, error_init
