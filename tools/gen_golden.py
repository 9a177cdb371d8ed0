#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the imported reference (CPU, this container only).

    python tools/gen_golden.py

Every fixture holds inputs and the reference's outputs for one hot-path function
(SURVEY.md §8a rows a5, a6, a7, a9, a10, a11).  Fixtures are data only; nothing from
/root/reference is copied.  Weights are the reference module's own default init under
torch.manual_seed(0) (no trained checkpoint exists offline).
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_import import import_reference  # noqa: E402
from adaptigraph_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF_CFG = "/root/reference/src/config"


def load_cfg(material):
    with open(f"{REF_CFG}/dynamics/{material}.yaml") as f:
        dyn = yaml.safe_load(f)
    with open(f"{REF_CFG}/planning/{material}.yaml") as f:
        plan = yaml.safe_load(f)["task_config"]
    return dyn, plan


def build_model(R, material, seed=0):
    dyn, _ = load_cfg(material)
    torch.manual_seed(seed)
    model = R.DynamicsPredictor(dyn["model_config"], dyn["material_config"], dyn["dataset_config"], "cpu")
    model.eval()
    return model


def onehots_to_edges(Rr, Rs):
    """(B, n_rel, N) one-hot pair -> per-sample count + padded index arrays (-1 pad)."""
    B, E, _ = Rr.shape
    valid = Rr.sum(-1) > 0
    n = valid.sum(1).numpy().astype(np.int32)
    recv = np.full((B, E), -1, np.int32)
    send = np.full((B, E), -1, np.int32)
    r = Rr.argmax(-1).numpy()
    s = Rs.argmax(-1).numpy()
    v = valid.numpy()
    recv[v] = r[v]
    send[v] = s[v]
    return n, recv, send


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


# ------------------------------------------------------------------ edge-builder fixtures (rows a6, a7)
def gen_edges(R):
    cases = []
    rng = np.random.default_rng(7)

    def add(name, material, n_obj, batch, variant, radius=None, n_pad=0, topk=None, connect=None, **kw):
        g = synth.make_graph_inputs(material, n_obj, batch, seed=len(cases) + 11, n_pad=n_pad, **kw)
        m = synth.MATERIALS[material]
        radius = m["radius"] if radius is None else radius
        topk = m["topk"] if topk is None else topk
        connect = m["connect_tools_all"] if connect is None else connect
        pos = g["state"][:, -1]
        if variant == "single":
            outs = [R.construct_edges_from_states(t(pos[b]), float(radius), t(g["mask"][b]), t(g["tool_mask"][b]),
                                                  topk=topk, connect_tools_all=connect) for b in range(batch)]
            E = max(o[0].shape[0] for o in outs)
            Rr = torch.zeros(batch, E, pos.shape[1])
            Rs = torch.zeros(batch, E, pos.shape[1])
            for b, (a, c) in enumerate(outs):
                Rr[b, :a.shape[0]] = a
                Rs[b, :c.shape[0]] = c
            rad_arr = np.full((batch,), radius, np.float64)
        else:
            if np.ndim(radius) == 0:
                thr = float(radius)
                rad_arr = np.full((batch,), radius, np.float64)
            else:
                thr = t(np.asarray(radius, np.float32))
                rad_arr = np.asarray(radius, np.float64)
            Rr, Rs = R.construct_edges_from_states_batch(t(pos), thr, t(g["mask"]), t(g["tool_mask"]),
                                                         topk=topk, connect_tools_all=connect)
        n, recv, send = onehots_to_edges(Rr, Rs)
        save("edges_" + name, pos=pos, mask=g["mask"], tool_mask=g["tool_mask"], radius=rad_arr,
             radius_is_tensor=np.array(np.ndim(radius) != 0), topk=np.int32(topk), connect_tools_all=np.array(connect),
             variant=np.array(variant), n_rel=n, recv=recv, send=send)
        cases.append(name)

    add("rope64_batch", "rope", 63, 3, "batch", spacing=0.1)
    add("rope64_single", "rope", 63, 2, "single", spacing=0.1)
    add("rope300_batch", "rope", 300, 2, "batch", spacing=0.2)
    add("rope300s01_batch", "rope", 300, 1, "batch", spacing=0.1)
    add("rope_thr04_batch", "rope", 120, 2, "batch", radius=0.4, spacing=0.1)      # fp32 r*r vs double r*r rounding
    add("rope_thr04_single", "rope", 120, 2, "single", radius=0.4, spacing=0.1)
    add("rope_persample_radius", "rope", 120, 3, "batch", radius=np.array([0.3, 0.5, 0.75], np.float32), spacing=0.1)
    add("rope_padded_batch", "rope", 50, 2, "batch", n_pad=14, spacing=0.1)
    add("rope_padded_single", "rope", 50, 2, "single", n_pad=14, spacing=0.1)
    add("rope_tiny_topk", "rope", 6, 2, "batch", topk=10, spacing=0.1)              # N < topk
    add("granular200_batch", "granular", 200, 2, "batch")
    add("granular200_single", "granular", 200, 1, "single")
    add("cloth256_batch", "cloth", 256, 2, "batch")
    add("cloth256_single", "cloth", 256, 2, "single")
    add("cloth64_far_batch", "cloth", 64, 2, "batch", tool_near=False)               # batch variant drops tool edges
    add("cloth64_far_single", "cloth", 64, 2, "single", tool_near=False)
    add("granular_connect_batch", "granular", 150, 2, "batch", connect=True)          # 5 tools + connect_tools_all
    add("granular_connect_single", "granular", 150, 2, "single", connect=True)
    del rng
    return cases


# ------------------------------------------------------------------ forward fixtures (row a5)
def gen_forward(R):
    def add(name, material, n_obj, batch, n_pad=0, decoder_scale=None, **kw):
        model = build_model(R, material)
        if decoder_scale is not None:
            with torch.no_grad():
                model.non_rigid_predictor.linear_2.weight *= decoder_scale
                model.non_rigid_predictor.linear_2.bias *= decoder_scale
        g = synth.make_graph_inputs(material, n_obj, batch, seed=3, n_pad=n_pad, **kw)
        m = synth.MATERIALS[material]
        Rr, Rs = R.construct_edges_from_states_batch(t(g["state"][:, -1]), float(m["radius"]), t(g["mask"]),
                                                     t(g["tool_mask"]), topk=m["topk"],
                                                     connect_tools_all=m["connect_tools_all"])
        graph = dict(state=t(g["state"]), attrs=t(g["attrs"]), action=t(g["action"]), p_instance=t(g["p_instance"]),
                     Rr=Rr, Rs=Rs)
        graph[material + "_physics_param"] = t(g["phys"])
        with torch.no_grad():
            pred_pos, pred_motion = model(**graph)
        n, recv, send = onehots_to_edges(Rr, Rs)
        save("fwd_" + name, material=np.array(material), state=g["state"], attrs=g["attrs"], action=g["action"],
             p_instance=g["p_instance"], phys=g["phys"], n_rel=n, recv=recv, send=send,
             decoder_scale=np.float32(1.0 if decoder_scale is None else decoder_scale),
             pred_pos=pred_pos.numpy(), pred_motion=pred_motion.numpy())

    add("rope64", "rope", 63, 2, spacing=0.1)
    add("rope301", "rope", 300, 1, spacing=0.2)                  # BASELINE configs[0]
    add("rope_padded", "rope", 50, 2, n_pad=14, spacing=0.1)
    add("rope_clamp", "rope", 63, 1, decoder_scale=5000.0, spacing=0.1)
    add("granular205", "granular", 200, 2)
    add("cloth257", "cloth", 256, 2)


# ------------------------------------------------------------------ rollout fixtures (rows a9, a10, a11)
def ppm_namespace(plan, device="cpu", radius=None):
    material = plan["material"]
    ns = types.SimpleNamespace()
    ns.task_config = plan
    ns.eef_num = plan["eef_num"]
    ns.material = material
    ns.material_dims = plan["material_dims"]
    ns.material_indices = plan["material_indices"]
    ns.adj_thresh = plan["adj_thresh"] if radius is None else radius
    ns.physics_param = {material: torch.tensor([0.5]).repeat(plan["material_dims"][material])}
    return ns


def gen_rollout(R):
    import contextlib
    import io

    def add(name, material, n_obj, bsz, n_look=1, len_lo=2, len_hi=6, **kw):
        _, plan = load_cfg(material)
        model = build_model(R, material)
        ppm = ppm_namespace(plan)
        state, act = synth.make_mpc_inputs(material, n_obj, bsz, n_look=n_look, seed=5, len_lo=len_lo, len_hi=len_hi, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            out = R.dynamics(t(state), t(act), model, "cpu", ppm)
        save("dyn_" + name, material=np.array(material), state=state, action=act,
             state_seqs=out["state_seqs"].numpy(), action_seqs=out["action_seqs"].numpy())

    add("rope60", "rope", 60, 6, spacing=0.1)
    add("rope60_look2", "rope", 60, 4, n_look=2, len_lo=1, len_hi=4, spacing=0.1)
    add("granular80", "granular", 80, 4, len_lo=1, len_hi=4)        # 5-point pusher
    add("cloth81", "cloth", 81, 4, len_lo=1, len_hi=4)               # connect_tools_all + gripper_enable

    def add_masked(name, material, n_obj, n_slots, bsz, **kw):
        _, plan = load_cfg(material)
        model = build_model(R, material)
        ppm = ppm_namespace(plan)
        rng = np.random.default_rng(9)
        state_init = np.zeros((bsz, n_slots, 3), np.float32)
        state_mask = np.zeros((bsz, n_slots), bool)
        acts = np.zeros((bsz, 4), np.float32)
        for b in range(bsz):
            nb = n_obj - 3 * b
            obj, a = synth.make_mpc_inputs(material, nb, 1, seed=20 + b, len_lo=1, len_hi=5, **kw)
            state_init[b, :nb] = obj
            state_mask[b, :nb] = True
            acts[b] = a[0, 0]
        del rng
        out = R.dynamics_masked(t(state_init), t(state_mask), t(acts), model, "cpu", ppm)
        save("dynmask_" + name, material=np.array(material), state_init=state_init, state_mask=state_mask,
             action=acts, state_seqs=out["state_seqs"].numpy(), action_seqs=out["action_seqs"].numpy())

    add_masked("rope40", "rope", 40, 44, 3, spacing=0.1)
    add_masked("granular60", "granular", 60, 64, 2)

    # decode_action (row a11): truncation toward zero of the length field, fp32 trig
    rng = np.random.default_rng(1)
    a = rng.uniform(-4, 15, (5, 3, 4)).astype(np.float32)
    for pl in (0.1, 0.2):
        d, r = R.decode_action(t(a), push_length=pl)
        save(f"decode_action_pl{int(pl * 10)}", action=a, push_length=np.float64(pl), decoded=d.numpy(), repeat=r.numpy())


def gen_weights(R):
    for material in ("rope", "granular", "cloth"):
        model = build_model(R, material)
        sd = {k: v.numpy() for k, v in model.state_dict().items()}
        assert len(sd) == 22 and sum(v.size for v in sd.values()) == 252903
        if material == "rope":
            save("weights_seed0", **sd)
        else:  # all three materials share dims, hence the same seed-0 init: keep one copy, assert the claim
            ref = np.load(os.path.join(OUT, "weights_seed0.npz"))
            assert all(np.array_equal(ref[k], sd[k]) for k in sd), material


# ------------------------------------------------------------------ MPPI glue ("next" row n1)
def gen_mppi(R):
    import contextlib
    import io
    L = R.losses
    for material, dyn in (("rope", "dyn_rope60"), ("rope", "dyn_rope60_look2"), ("granular", "dyn_granular80"), ("cloth", "dyn_cloth81")):
        g = np.load(os.path.join(OUT, dyn + ".npz"))
        _, plan = load_cfg(material)
        ratio = plan["sim_real_ratio"]
        state_seqs, action, state_cur = t(g["state_seqs"]), t(g["action"]), t(g["state"])
        bsz, nl, n, _ = state_seqs.shape
        rng = np.random.default_rng(31)
        target = (g["state"][rng.choice(n, 40, replace=False)] + rng.normal(0, 0.05, (40, 3))).astype(np.float32)
        lo, hi = g["state"].min(0), g["state"].max(0)
        box = np.array([[lo[0] + 0.2, hi[0] - 0.1], [lo[2] - 0.1, hi[2] + 0.3]], np.float32)
        bbox = np.array([[lo[0] - 0.5, hi[0] + 0.5], [lo[2] - 0.5, hi[2] + 0.5]], np.float64)
        flat = state_seqs.reshape(bsz * nl, n, 3)
        out = dict(material=np.array(material), state_seqs=g["state_seqs"], action=g["action"], state_cur=g["state"],
                   target=target, box=box, bbox=bbox, sim_real_ratio=np.float64(ratio),
                   chamfer=L.chamfer(flat, t(target)[None]).numpy(), box_loss=L.box_loss(flat, t(box)).numpy())
        pen = {"rope": L.rope_penalty, "granular": L.granular_penalty, "cloth": L.cloth_penalty}[material]
        out["penalty"] = pen(state_seqs, action, state_cur, sim_real_ratio=ratio).numpy()
        from functools import partial
        for crit_name, crit in (("chamfer", partial(L.chamfer, y=t(target)[None])), ("box", partial(L.box_loss, target=t(box)))):
            with contextlib.redirect_stdout(io.StringIO()):
                r = R.running_cost(state_seqs, action, state_cur, error_func=crit,
                                   penalty_func=partial(pen, sim_real_ratio=ratio), bbox=bbox)
            out["reward_" + crit_name] = r["reward_seqs"].numpy()
        # MPPI update on these rewards (plan_utils.py:80-101) with the task's limits / weights
        lim_lo, lim_hi = t(np.array(plan["action_lower_lim"], np.float32)), t(np.array(plan["action_upper_lim"], np.float32))
        upd = R.plan_utils.optimize_action_mppi(action, t(out["reward_chamfer"]), reward_weight=plan["reward_weight"],
                                                action_lower_lim=lim_lo, action_upper_lim=lim_hi, push_length=plan["push_length"])
        out.update(lim_lo=lim_lo.numpy(), lim_hi=lim_hi.numpy(), reward_weight=np.float64(plan["reward_weight"]),
                   push_length=np.float64(plan["push_length"]), mppi_act_seq=upd.numpy())
        save("mppi_" + dyn[4:], **out)

    # action sampling (plan_utils.py:35-77): CPU torch RNG under a fixed seed is part of the contract we mirror
    _, plan = load_cfg("rope")
    lim_lo, lim_hi = t(np.array(plan["action_lower_lim"], np.float32)), t(np.array(plan["action_upper_lim"], np.float32))
    act_seq = t(np.array([[-2.0, 1.0, 0.5, 8.0], [-1.0, 0.5, -1.0, 6.0]], np.float32))
    outs = {}
    for it in (0, 1):
        torch.manual_seed(1234)
        outs[f"samples_it{it}"] = R.plan_utils.sample_action_seq(act_seq, lim_lo, lim_hi, 16, "cpu", iter_index=it,
                                                                  noise_level=plan["noise_level"], push_length=plan["push_length"]).numpy()
    raw = t(np.array([[1.0, 9.0, 4.0, 20.0], [-9.0, -9.0, -7.0, 1.0], [-1.0, 1.0, 3.2, 7.0]], np.float32))
    save("mppi_sampling", act_seq=act_seq.numpy(), lim_lo=lim_lo.numpy(), lim_hi=lim_hi.numpy(), seed=np.int64(1234),
         noise_level=np.float64(plan["noise_level"]), push_length=np.float64(plan["push_length"]),
         clip_in=raw.numpy(), clip_out=R.plan_utils.clip_actions(raw, lim_lo, lim_hi).numpy(), **outs)


# ------------------------------------------------------------------ Planner(config) (row n1: planner.py:38-326, MPPI branch)
def toy_rollout(state_cur, act_seqs):
    """An analytic stand-in for model_rollout_fn (the SAME function is restated in tests/test_mppi.py): every particle is
    displaced by a smooth function of the action, accumulated over the look-ahead steps."""
    n, L = act_seqs.shape[0], act_seqs.shape[1]
    disp = torch.stack([torch.sin(act_seqs[..., 0]) * act_seqs[..., 3], 0.1 * act_seqs[..., 2], torch.cos(act_seqs[..., 1])], -1)  # (n, L, 3)
    return {"state_seqs": state_cur[None, None] + 0.05 * torch.cumsum(disp, 1)[:, :, None, :] * torch.ones(n, L, state_cur.shape[0], 1)}


def toy_cost(state_seqs, act_seqs, state_cur=None, weights=None, target=None):
    return {"reward_seqs": -((state_seqs[:, -1] - target[None]) ** 2).sum((1, 2)) - 0.01 * (act_seqs ** 2).sum((1, 2))}


def gen_planner(R):
    import contextlib
    import io
    from functools import partial
    rng = np.random.default_rng(77)
    state_cur = t(rng.normal(0, 1, (12, 3)).astype(np.float32))
    target = state_cur + t(np.array([0.3, 0.0, -0.2], np.float32))
    lo, hi = t(np.array([-3.0, -3.0, -3.14, 1.0], np.float32)), t(np.array([3.0, 3.0, 3.14, 6.0], np.float32))
    act0 = t(np.array([[0.5, -0.5, 0.3, 3.0], [1.0, 0.2, -0.4, 2.0]], np.float32))
    cfg = dict(action_dim=4, model_rollout_fn=toy_rollout, evaluate_traj_fn=partial(toy_cost, target=target), n_sample=32, n_look_ahead=2,
               n_update_iter=3, reward_weight=20.0, action_lower_lim=lo, action_upper_lim=hi, planner_type="MPPI", device="cpu",
               noise_level=0.4)
    out = dict(state_cur=state_cur.numpy(), target=target.numpy(), lim_lo=lo.numpy(), lim_hi=hi.numpy(), act0=act0.numpy(),
               n_sample=np.int64(32), n_update_iter=np.int64(3), reward_weight=np.float64(20.0), noise_level=np.float64(0.4), seed=np.int64(99))
    res_list = []
    torch.manual_seed(99)
    holder = []
    # the reference calls its sampler with iter_index=... (planner.py:243), which its own default sampler does not accept: as in
    # plan.py a sampling_action_seq_fn is always supplied — here a wrapper around that default sampler
    cfg["sampling_action_seq_fn"] = lambda act_seq, iter_index=0: holder[-1].sample_action_sequences_default(act_seq)
    for c in range(2):                                   # two chunks, then the reference's merge_res (plan.py chunk loop)
        planner = R.Planner(cfg)
        holder.append(planner)
        with contextlib.redirect_stdout(io.StringIO()):
            res = planner.trajectory_optimization(state_cur, act0.clone())
        res_list.append(res)
        out[f"chunk{c}_act_seq"] = res["act_seq"].numpy()
        out[f"chunk{c}_best_reward"] = res["best_eval_output"]["reward_seqs"].numpy()
        out[f"chunk{c}_best_states"] = res["best_model_output"]["state_seqs"].numpy()
    merged = planner.merge_res(res_list)
    out["merged_act_seq"] = merged["act_seq"].numpy()
    save("planner_mppi_toy", **out)


# ------------------------------------------------------------------ sys-id objective ("next" row n2)
def gen_sysid(R):
    P = R.physics_param_optimizer
    for material, n0, kw in (("rope", 40, dict(spacing=0.1)), ("granular", 60, {})):
        _, plan = load_cfg(material)
        plan = dict(plan, max_nobj=72)                 # keep the CPU reference run and the fixture small
        model = build_model(R, material)
        ppm = ppm_namespace(plan)
        ppm.model, ppm.device = model, "cpu"
        rng = np.random.default_rng(77)
        inits, reals, acts = [], [], []
        for b in range(3):
            nb = n0 - 4 * b
            obj, a = synth.make_mpc_inputs(material, nb, 1, seed=40 + b, len_lo=1, len_hi=4, **kw)
            inits.append(obj.astype(np.float32))
            keep = np.sort(rng.choice(nb, nb - 1 - b, replace=False))      # the "observed" cloud has a different point count
            reals.append((obj[keep] + rng.normal(0, 0.03, (len(keep), 3))).astype(np.float32))
            acts.append(a[0, 0])
        out = {}
        for i, pv in enumerate((0.2, 0.5, 0.9)):
            out[f"error_p{i}"] = np.float64(P.dynamics_error([pv], ppm, inits, reals, acts))
        # per-sample values behind the mean at phys 0.5 (mean_chamfer on the padded tensors)
        mx = plan["max_nobj"]
        pad = lambda L: np.stack([np.pad(x, ((0, mx - len(x)), (0, 0))) for x in L])
        msk = lambda L: np.stack([np.arange(mx) < len(x) for x in L])
        o = R.dynamics_masked(t(pad(inits)), t(msk(inits)), t(np.stack(acts)), model, "cpu", ppm,
                              physics_param={material: torch.tensor([0.5])})
        per = R.losses.mean_chamfer(o["state_seqs"].detach(), t(pad(reals)), t(msk(inits)), t(msk(reals)))
        save("sysid_" + material, material=np.array(material), max_nobj=np.int64(mx), phys=np.array([0.2, 0.5, 0.9]),
             n_init=np.array([len(x) for x in inits]), n_real=np.array([len(x) for x in reals]),
             state_init=pad(inits), state_real=pad(reals), action=np.stack(acts), state_pred_p1=o["state_seqs"].numpy(),
             chamfer_p1=np.asarray(per, np.float64), **out)


# ------------------------------------------------------------------ eval rollout over a dataset ("next" row n3)
def write_eval_dataset(root, g):
    """Lay a fixture (arrays in an npz / dict) out in the reference's on-disk dataset format under `root`."""
    import pickle
    name = str(g["data_name"])
    n_epi = len(g["n_frames"])
    prep = os.path.join(root, "preprocess", name)
    os.makedirs(os.path.join(prep, "frame_pairs"), exist_ok=True)
    eef, obj = [], []
    for e in range(n_epi):
        os.makedirs(os.path.join(root, "sim_data", name, f"{e:06}"), exist_ok=True)
        with open(os.path.join(root, "sim_data", name, f"{e:06}", "property_params.pkl"), "wb") as f:
            pickle.dump({"particle_radius": 0.03, "stiffness": float(g["stiffness"][e])}, f)
        T = int(g["n_frames"][e])
        eef.append(np.asarray(g["eef_pos"][e, :T]))
        obj.append(np.asarray(g["obj_pos"][e, :T]))
        for k in range(int(g["n_push"][e])):
            np.savetxt(os.path.join(prep, "frame_pairs", f"{e:06}_{k + 1:02}.txt"), g[f"pairs_{e}_{k + 1}"], fmt="%d")
    with open(os.path.join(prep, "positions.pkl"), "wb") as f:
        pickle.dump({"eef_pos": eef, "obj_pos": obj}, f)


def gen_evalrollout(R):
    import contextlib
    import importlib
    import io
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "adaptigraph_amd"))
    from adaptigraph_amd import sampling
    dgraph = importlib.import_module("dynamics.dataset.graph")
    # DGL is absent: stage 1 of fps() runs OUR restatement of dgl.geometry.farthest_point_sampler (tensor in, tensor out)
    dgraph.farthest_point_sampler = lambda x, n, start_idx=0: torch.from_numpy(sampling.farthest_point_sampler(x.numpy(), n, start_idx))
    rr = importlib.import_module("dynamics.rollout.rollout")
    rg = importlib.import_module("dynamics.rollout.graph")
    prep = importlib.import_module("dynamics.preprocess.preprocess")
    dutils = importlib.import_module("dynamics.utils")

    # fps_rad_idx straight from the reference
    rng = np.random.default_rng(5)
    cloud = rng.uniform(0, 2, (150, 3)).astype(np.float32)
    np.random.seed(11)
    _, rad_idx = dutils.fps_rad_idx(cloud, 0.45)
    np.random.seed(12)
    fps_idx = dgraph.fps(cloud, 60, [0.3, 0.5])
    save("fps_cloud", cloud=cloud, radius=np.float64(0.45), seed=np.int64(11), rad_idx=rad_idx.astype(np.int64),
         fps_seed=np.int64(12), fps_max_nobj=np.int64(60), fps_range=np.array([0.3, 0.5]), fps_idx=fps_idx.astype(np.int64))

    # synthetic rope dataset: 3 episodes x 2 pushes, tool sweeping through a 120-particle rope
    dyn, _ = load_cfg("rope")
    n_his, n_future = dyn["dataset_config"]["n_his"], dyn["dataset_config"]["n_future"]
    n_part, T_push = 120, 12
    fx = dict(data_name=np.array("rope"), n_frames=[], n_push=[], stiffness=[])
    eef_all, obj_all = [], []
    for e in range(3):
        r = np.random.default_rng(100 + e)
        i = np.arange(n_part)
        obj0 = np.stack([i * 0.04, np.zeros(n_part), 1.2 * np.sin(2 * np.pi * i / n_part)], 1) + r.normal(0, 0.01, (n_part, 3))
        eef_e, obj_e, n_frames = [], [], 0
        cur = obj0.copy()
        for k in range(2):
            a = r.uniform(0, 2 * np.pi)
            p0 = np.array([r.uniform(1.0, 3.8), 0.0, r.uniform(-1.0, 1.0)])
            step = 0.06 * np.array([np.cos(a), 0.0, np.sin(a)])
            eef_k, obj_k = [], []
            for tt_ in range(T_push):
                tool = p0 + step * tt_
                d = np.linalg.norm(cur[:, [0, 2]] - tool[[0, 2]], axis=1)
                cur = cur + np.exp(-(d / 0.3) ** 2)[:, None] * step * 0.8 + r.normal(0, 0.002, cur.shape)
                eef_k.append(tool[None].copy()); obj_k.append(cur.copy())
            eef_k, obj_k = np.array(eef_k), np.array(obj_k)
            with contextlib.redirect_stdout(io.StringIO()):
                pairs, cnt = prep.extract_push(eef_k, dyn["dataset_config"]["dist_thresh"], n_his, n_future, n_frames)
            n_frames += cnt
            fx[f"pairs_{e}_{k + 1}"] = np.asarray(pairs, np.int64)
            eef_e.append(eef_k); obj_e.append(obj_k)
        fx["n_frames"].append(n_frames); fx["n_push"].append(2); fx["stiffness"].append(r.uniform(0.1, 0.9))
        eef_all.append(np.concatenate(eef_e).astype(np.float32)); obj_all.append(np.concatenate(obj_e).astype(np.float32))
    fx["eef_pos"], fx["obj_pos"] = np.stack(eef_all), np.stack(obj_all)
    fx = {k: np.asarray(v) for k, v in fx.items()}

    with tempfile.TemporaryDirectory() as root:
        write_eval_dataset(root, fx)
        ds = dict(dyn["dataset_config"], data_dir=os.path.join(root, "sim_data"), prep_data_dir=os.path.join(root, "preprocess"),
                  device="cpu", ratio={"train": [0, 0.34], "valid": [0.34, 1.0]})
        ds["datasets"] = [dict(ds["datasets"][0], max_nobj=40, max_nR=400)]
        config = dict(dyn, dataset_config=ds)
        model = build_model(R, "rope")
        # pieces, for the loaders / graph set-up tests
        pair_lists, phys = R_load(ds, dyn["material_config"])
        eef_pos, obj_pos = importlib.import_module("dynamics.dataset.load").load_positions(ds)
        np.random.seed(3)
        pair0 = fx["pairs_1_1"][0]
        graph, fidx = rg.construct_graph(ds, dyn["material_config"], eef_pos[1], obj_pos[1], n_his, pair0, phys[1])
        pairs_e1 = pair_lists[pair_lists[:, 0] == 1][:, 1:]
        sched, cur = [(int(pair0[n_his - 1]), int(pair0[n_his]))], int(pair0[n_his])
        while True:
            nxt = rg.get_next_pair_or_break_episode_pushes(pairs_e1, n_his, obj_pos[1].shape[0], cur)
            if nxt is None:
                break
            sched.append((int(nxt[n_his - 1]), int(nxt[n_his]))); cur = int(nxt[n_his])
        nxt_skip = rg.get_next_pair_or_break_episode(pairs_e1[pairs_e1[:, n_his - 1] % 3 == 0], n_his, obj_pos[1].shape[0], 1)
        # the whole driver
        out_dir = os.path.join(root, "out")
        os.makedirs(out_dir)
        np.random.seed(42)
        with contextlib.redirect_stdout(io.StringIO()):
            rr.rollout_dataset(model, "cpu", config, out_dir, False)
        errs = {}
        for e in (1, 2):
            for k in (1, 2):
                errs[f"error_{e}_{k}"] = np.loadtxt(os.path.join(out_dir, f"{e}", "short", f"error_{k}.txt"))
        save("evalrollout_rope", **fx, seed=np.int64(42), max_nobj=np.int64(40), max_nR=np.int64(400),
             pair_lists=pair_lists.astype(np.int64), phys_norm=np.array([p["rope"] for p in phys], np.float32),
             graph_seed=np.int64(3), graph_pair=pair0, graph_fps_idx=np.asarray(fidx, np.int64),
             **{"graph_" + k: v.numpy() for k, v in graph.items()}, schedule=np.array(sched, np.int64),
             next_skip=np.asarray(nxt_skip, np.int64), error_short=np.loadtxt(os.path.join(out_dir, "error_short.txt")), **errs)


# ------------------------------------------------------------------ training unroll ("next" row n4)
def gen_train(R):
    import contextlib
    import importlib
    import io
    import tempfile
    from adaptigraph_amd import sampling
    dgraph = importlib.import_module("dynamics.dataset.graph")
    dgraph.farthest_point_sampler = lambda x, n, start_idx=0: torch.from_numpy(sampling.farthest_point_sampler(x.numpy(), n, start_idx))
    dds = importlib.import_module("dynamics.dataset.dataset")
    fx = dict(np.load(os.path.join(OUT, "evalrollout_rope.npz")))      # the synthetic on-disk dataset of row n3
    dyn, _ = load_cfg("rope")
    n_future = dyn["dataset_config"]["n_future"]
    with tempfile.TemporaryDirectory() as root:
        write_eval_dataset(root, fx)
        ds = dict(dyn["dataset_config"], data_dir=os.path.join(root, "sim_data"), prep_data_dir=os.path.join(root, "preprocess"),
                  device="cpu", ratio={"train": [0, 0.67], "valid": [0.67, 1.0]}, verbose=False)
        ds["datasets"] = [dict(ds["datasets"][0], max_nobj=40, max_nR=400)]
        with contextlib.redirect_stdout(io.StringIO()):
            dset = dds.DynDataset(ds, dyn["material_config"], phase="train")
        out = dict(seed=np.int64(7), idx=np.array([3, 20, 41], np.int64), n_samples=np.int64(len(dset)),
                   state_noise=np.float64(ds["randomness"]["state_noise"]["train"]))
        samples = []
        for k, i in enumerate(out["idx"]):
            np.random.seed(int(out["seed"]) + k)
            samples.append(dset[int(i)])
        keys = ["state", "action", "eef_future", "action_future", "state_future", "attrs", "p_instance", "obj_mask", "rope_physics_param"]
        batch = {k: torch.stack([s[k] for s in samples]) for k in keys + ["Rr", "Rs", "p_rigid", "material_index"]}
        for k in keys:
            out["b_" + k] = batch[k].numpy()
        # edges as lists (dense Rr/Rs would be 3 x 400 x 41 x 2)
        n_rel = [int((s["Rr"].sum(1) > 0).sum()) for s in samples]
        cap = max(n_rel)
        recv, send = np.zeros((3, cap), np.int32), np.zeros((3, cap), np.int32)
        for b, s in enumerate(samples):
            recv[b, :n_rel[b]] = s["Rr"][:n_rel[b]].argmax(1).numpy()
            send[b, :n_rel[b]] = s["Rs"][:n_rel[b]].argmax(1).numpy()
        out.update(n_rel=np.array(n_rel, np.int32), recv=recv, send=send)
        # the objective of train.py:84-108 and its gradients, on the reference model with seed-0 weights
        model = build_model(R, "rope")
        model.train()
        data = {k: v.clone() for k, v in batch.items()}
        mse = torch.nn.MSELoss()
        loss_sum, preds = 0, []
        for fi in range(n_future):
            gt_state = data["state_future"][:, fi].clone()
            pred_state, pred_motion = model(**data)
            pred_state_p = pred_state[:, :gt_state.shape[1], :3].clone()
            preds.append(pred_state_p.detach().numpy())
            loss_sum = loss_sum + mse(pred_state_p, gt_state)
            if fi < n_future - 1:
                next_eef = data["eef_future"][:, fi].clone()
                next_action = data["action_future"][:, fi].clone()
                next_state = next_eef.unsqueeze(1)
                next_state[:, -1, :pred_state_p.shape[1]] = pred_state_p
                data["state"] = torch.cat([data["state"][:, 1:], next_state], dim=1)
                data["action"] = next_action
        loss_sum.backward()
        out["loss"] = np.float64(loss_sum.item())
        out["preds"] = np.stack(preds)
        for k, p in model.named_parameters():
            out["grad_" + k] = p.grad.numpy()
        # one Adam step as train.py does it (lr 1e-3): the updated first-layer weights pin the optimiser settings
        opt = torch.optim.Adam(model.parameters(), lr=0.001)
        opt.step()
        out["adam_particle_encoder.model.0.weight"] = model.state_dict()["particle_encoder.model.0.weight"].numpy()
        save("train_rope", **out)


def R_load(ds, material_config):
    import contextlib
    import importlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return importlib.import_module("dynamics.dataset.load").load_dataset(ds, material_config, phase="valid")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    R = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "mppi":
        return gen_mppi(R)
    if len(sys.argv) > 1 and sys.argv[1] == "planner":
        return gen_planner(R)
    if len(sys.argv) > 1 and sys.argv[1] == "sysid":
        return gen_sysid(R)
    if len(sys.argv) > 1 and sys.argv[1] == "evalrollout":
        return gen_evalrollout(R)
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        return gen_train(R)
    gen_weights(R)
    gen_edges(R)
    gen_forward(R)
    gen_rollout(R)
    gen_mppi(R)
    gen_planner(R)
    gen_sysid(R)
    gen_evalrollout(R)
    gen_train(R)


if __name__ == "__main__":
    main()
