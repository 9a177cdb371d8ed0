"""MPPI glue ("next" row n1): oracle and host mirror pinned to goldens from the reference (CPU), HIP cost kernel and
the planner step against the oracle (GPU)."""
from functools import partial

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from adaptigraph_amd import configs, losses, mpc, synth
from adaptigraph_amd import plan_utils as pu
from oracle import ag_oracle as ago

CASES = [n for n in golden_files("mppi_") if n != "mppi_sampling"]
PEN_T = {"rope": losses.rope_penalty, "granular": losses.granular_penalty, "cloth": losses.cloth_penalty}
PEN_O = {"rope": ago.rope_penalty, "granular": ago.granular_penalty, "cloth": ago.cloth_penalty}


@pytest.mark.parametrize("name", CASES)
def test_oracle_cost_terms_match_reference(name):
    g = load_golden(name)
    mat, ratio = str(g["material"]), float(g["sim_real_ratio"])
    flat = g["state_seqs"].reshape(-1, g["state_seqs"].shape[2], 3)
    assert np.abs(ago.chamfer(flat, g["target"][None]) - g["chamfer"]).max() <= 2e-6
    assert np.abs(ago.box_loss(flat, g["box"]) - g["box_loss"]).max() <= 2e-6
    assert np.abs(PEN_O[mat](g["state_seqs"], g["action"], g["state_cur"], ratio) - g["penalty"]).max() <= 2e-6
    pen = partial(PEN_O[mat], sim_real_ratio=ratio)
    for crit, ef in (("chamfer", partial(ago.chamfer, y=g["target"][None])), ("box", partial(ago.box_loss, target=g["box"]))):
        r = ago.running_cost(g["state_seqs"], g["action"], g["state_cur"], ef, pen, g["bbox"])
        assert np.abs(r - g["reward_" + crit]).max() <= 2e-5
    upd = ago.optimize_action_mppi(g["action"], g["reward_chamfer"], float(g["reward_weight"]), g["lim_lo"], g["lim_hi"],
                                   float(g["push_length"]))
    assert np.abs(upd - g["mppi_act_seq"]).max() <= 2e-5


@pytest.mark.parametrize("name", CASES)
def test_host_mirror_matches_reference_on_cpu(name):
    """The torch-op parts of the mirror (penalties, box loss, MPPI update, running_cost with the box criterion) are
    device-agnostic; on CPU they reproduce the reference bit for bit."""
    g = load_golden(name)
    mat, ratio = str(g["material"]), float(g["sim_real_ratio"])
    st, ac, sc = (torch.from_numpy(g[k]) for k in ("state_seqs", "action", "state_cur"))
    assert np.array_equal(PEN_T[mat](st, ac, sc, sim_real_ratio=ratio).numpy(), g["penalty"])
    assert np.array_equal(losses.box_loss(st.reshape(-1, st.shape[2], 3), torch.from_numpy(g["box"])).numpy(), g["box_loss"])
    r = mpc.running_cost(st, ac, sc, error_func=partial(losses.box_loss, target=torch.from_numpy(g["box"])),
                         penalty_func=partial(PEN_T[mat], sim_real_ratio=ratio), bbox=g["bbox"])["reward_seqs"]
    assert np.abs(r.numpy() - g["reward_box"]).max() <= 1e-6
    upd = pu.optimize_action_mppi(ac, torch.from_numpy(g["reward_chamfer"]), reward_weight=float(g["reward_weight"]),
                                  action_lower_lim=torch.from_numpy(g["lim_lo"]), action_upper_lim=torch.from_numpy(g["lim_hi"]),
                                  push_length=float(g["push_length"]))
    assert np.abs(upd.numpy() - g["mppi_act_seq"]).max() <= 1e-6


def test_action_sampling_consumes_rng_like_the_reference():
    g = load_golden("mppi_sampling")
    lo, hi = torch.from_numpy(g["lim_lo"]), torch.from_numpy(g["lim_hi"])
    assert np.array_equal(pu.clip_actions(torch.from_numpy(g["clip_in"]), lo, hi).numpy(), g["clip_out"])
    assert np.abs(ago.clip_actions(g["clip_in"], g["lim_lo"], g["lim_hi"]) - g["clip_out"]).max() <= 1e-6
    for it in (0, 1):
        torch.manual_seed(int(g["seed"]))
        s = pu.sample_action_seq(torch.from_numpy(g["act_seq"]), lo, hi, 16, "cpu", iter_index=it,
                                 noise_level=float(g["noise_level"]), push_length=float(g["push_length"]))
        assert np.array_equal(s.numpy(), g[f"samples_it{it}"])
        assert np.array_equal(s[0].numpy(), g["act_seq"]) or it == 0      # sample 0 is the unperturbed sequence


def test_chamfer_refuses_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.chamfer(torch.zeros(2, 5, 3), torch.zeros(1, 4, 3))


# ------------------------------------------------------------------------------------------------------ GPU
DEV = "cuda:0"


def tg(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_chamfer_kernel_and_running_cost_golden(name):
    g = load_golden(name)
    mat, ratio = str(g["material"]), float(g["sim_real_ratio"])
    st, ac, sc = tg(g["state_seqs"]), tg(g["action"]), tg(g["state_cur"])
    flat = st.reshape(-1, st.shape[2], 3)
    assert np.abs(losses.chamfer(flat, tg(g["target"])[None]).cpu().numpy() - g["chamfer"]).max() <= 2e-6
    r = mpc.running_cost(st, ac, sc, error_func=partial(losses.chamfer, y=tg(g["target"])[None]),
                         penalty_func=partial(PEN_T[mat], sim_real_ratio=ratio), bbox=g["bbox"])["reward_seqs"]
    assert np.abs(r.cpu().numpy() - g["reward_chamfer"]).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,M,batched", [(1024, 1000, 1000, False), (7, 4096, 333, True), (3, 1, 1, False), (5, 37, 2050, True)])
def test_chamfer_kernel_vs_oracle_sizes(B, N, M, batched):
    rng = np.random.default_rng(B + N)
    x = rng.normal(0, 2, (B, N, 3)).astype(np.float32)
    y = rng.normal(0.3, 2, (B if batched else 1, M, 3)).astype(np.float32)
    got = losses.chamfer(tg(x), tg(y)).cpu().numpy()
    sub = slice(0, min(B, 16))                      # the oracle materialises (B,M,N): check a slice at the big size
    ref = ago.chamfer(x[sub], y[sub] if batched else y)
    assert np.abs(got[sub] - ref).max() <= 5e-6 * max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all()


@pytest.mark.gpu
def test_chamfer_at_the_lds_limit_with_odd_cloud_sizes():
    """N + M = 12 800 points is the documented limit of ag_chamfer (both clouds resident in LDS); r06 keeps the clouds as structure-of-arrays planes padded to
    an even number of points (+inf pad points, read as pairs): odd sizes at the limit, one point beyond it is refused with an error code, not a crash."""
    rng = np.random.default_rng(5)
    x = rng.normal(0, 2, (1, 6401, 3)).astype(np.float32)
    y = rng.normal(0.3, 2, (1, 6399, 3)).astype(np.float32)
    got = losses.chamfer(tg(x), tg(y)).cpu().numpy()
    ref = ago.chamfer(x, y)
    assert np.abs(got - ref).max() <= 5e-6 * max(1.0, float(np.abs(ref).max()))
    with pytest.raises(RuntimeError, match="LDS-resident limit"):
        losses.chamfer(tg(x), tg(np.concatenate([y, y[:, :1]], 1)))


@pytest.mark.gpu
def test_mppi_step_vs_oracle_chain(weights):
    """One planner update from given samples: engine rollout + HIP chamfer + penalties + softmax update, against the
    oracle's rollout -> cost -> update chain on the same samples (exact-fp32 engine mode: no top-k flips)."""
    from adaptigraph_amd.model import DynamicsPredictor
    mat = "rope"
    task = configs.task_config(mat)
    state, act = synth.make_mpc_inputs(mat, 120, 24, seed=17, len_lo=2, len_hi=4.9, spacing=0.1)
    target = (state[::3] + np.array([0.3, 0.0, 0.2], np.float32)).astype(np.float32)
    bbox = np.array([[state[:, 0].min() - 1, state[:, 0].max() + 1], [state[:, 2].min() - 1, state[:, 2].max() + 1]])
    lo, hi = np.array(task["action_lower_lim"], np.float32), np.array(task["action_upper_lim"], np.float32)
    model = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), DEV)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    model = model.to(DEV).eval().set_option("precision", 0)
    ppm = configs.ppm_optimizer_stub(mat)
    ppm.physics_param = {mat: torch.tensor([0.5], device=DEV)}
    planner = mpc.MPPIPlanner(model, DEV, ppm, partial(losses.chamfer, y=tg(target)[None]),
                              partial(losses.rope_penalty, sim_real_ratio=task["sim_real_ratio"]), bbox, lo, hi, n_sample=24,
                              reward_weight=500.0, noise_level=1.0)
    new_seq, reward, out = planner.step(tg(state), tg(act))
    seq_ref, _ = ago.dynamics(weights, task, state, act)
    r_ref = ago.running_cost(seq_ref, act, state, partial(ago.chamfer, y=target[None]),
                             partial(ago.rope_penalty, sim_real_ratio=task["sim_real_ratio"]), bbox)
    upd_ref = ago.optimize_action_mppi(act, r_ref, 500.0, lo, hi, task["push_length"])
    assert np.abs(out["state_seqs"].cpu().numpy() - seq_ref).max() <= 1e-4
    assert np.abs(reward.cpu().numpy() - r_ref).max() <= 1e-4
    assert np.abs(new_seq.cpu().numpy() - upd_ref).max() <= 5e-3     # softmax(500 * reward) amplifies 1e-5 reward noise
    res = planner.trajectory_optimization_mppi(tg(state), tg(act[0]))
    assert res["act_seq"].shape == (1, 4) and torch.isfinite(res["best_reward"])


# ------------------------------------------------------------------ Planner(config) drop-in (planner.py:38-326, MPPI branch)
def _toy_rollout(state_cur, act_seqs):
    """The analytic model_rollout_fn the golden was generated with (tools/gen_golden.py:toy_rollout)."""
    n, L = act_seqs.shape[0], act_seqs.shape[1]
    disp = torch.stack([torch.sin(act_seqs[..., 0]) * act_seqs[..., 3], 0.1 * act_seqs[..., 2], torch.cos(act_seqs[..., 1])], -1)
    return {"state_seqs": state_cur[None, None] + 0.05 * torch.cumsum(disp, 1)[:, :, None, :] * torch.ones(n, L, state_cur.shape[0], 1)}


def _toy_cost(state_seqs, act_seqs, state_cur=None, weights=None, target=None):
    return {"reward_seqs": -((state_seqs[:, -1] - target[None]) ** 2).sum((1, 2)) - 0.01 * (act_seqs ** 2).sum((1, 2))}


def test_planner_config_dropin_matches_reference_planner():
    """`mpc.Planner(config)` with the reference's config keys, its clip / softmax update, three MPPI iterations and the two-chunk
    merge, against the reference Planner run on the same callbacks under the same torch seed (CPU: the sampler's RNG draw order is
    part of the contract).  As in plan.py a sampling_action_seq_fn is supplied (the reference's own default sampler does not take
    the iter_index it is called with, planner.py:243): here a wrapper around each class's default sampler."""
    g = load_golden("planner_mppi_toy")
    tt = lambda k: torch.from_numpy(g[k].copy())
    cfg = dict(action_dim=4, model_rollout_fn=_toy_rollout, evaluate_traj_fn=partial(_toy_cost, target=tt("target")), n_sample=int(g["n_sample"]),
               n_look_ahead=2, n_update_iter=int(g["n_update_iter"]), reward_weight=float(g["reward_weight"]), action_lower_lim=tt("lim_lo"),
               action_upper_lim=tt("lim_hi"), planner_type="MPPI", device="cpu", noise_level=float(g["noise_level"]))
    torch.manual_seed(int(g["seed"]))
    res, holder = [], []
    cfg["sampling_action_seq_fn"] = lambda act_seq, iter_index=0: holder[-1].sample_action_sequences_default(act_seq)
    for c in range(2):
        planner = mpc.Planner(cfg)
        holder.append(planner)
        r = planner.trajectory_optimization(tt("state_cur"), tt("act0"))
        assert set(r) == {"act_seq", "model_outputs", "eval_outputs", "best_model_output", "best_eval_output"}
        assert np.abs(r["act_seq"].numpy() - g[f"chunk{c}_act_seq"]).max() <= 1e-6
        assert np.abs(r["best_eval_output"]["reward_seqs"].numpy() - g[f"chunk{c}_best_reward"]).max() <= 1e-5
        assert np.abs(r["best_model_output"]["state_seqs"].numpy() - g[f"chunk{c}_best_states"]).max() <= 1e-6
        res.append(r)
    assert np.abs(planner.merge_res(res)["act_seq"].numpy() - g["merged_act_seq"]).max() <= 1e-6
    with pytest.raises(NotImplementedError):
        mpc.Planner(dict(cfg, planner_type="GD")).trajectory_optimization(tt("state_cur"), tt("act0"))
