#!/usr/bin/env python3
"""Stage-0 probe for a shared-state rollout: how much of BASELINE configs[4] repeats one base trajectory?

dynamics() takes ONE `state` for `bsz` action samples (forward_dynamics.py:11-38): until a sample's tool has touched a particle — and,
afterwards, outside the 3-hops-per-step light cone of the touched particles — the sample's predictions are those of a rollout in which the
tool does nothing.  This script measures that with today's engine: it runs the MPPI shape (rope-1000 + tool, `--samples` pushes, `--steps` model
steps) once with the sampled pushes and once with every tool slot masked invalid, one rollout per step count t = 1..T (the engine returns
the state after its last step), and counts per step the (sample, particle) pairs whose position is BIT-EQUAL to the tool-less run, plus the
particles within the interaction radius of a clean/dirty boundary.  No engine change; output -> profiles/r06_shared_state_probe.txt.

    python tools/shared_state_probe.py [--samples 1024] [--steps 15] [--material rope] [--particles 1000]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from adaptigraph_amd import _lib, configs, synth                               # noqa: E402
from adaptigraph_amd.forward_dynamics import _place_tool_lean, _constants, rollout   # noqa: E402
from adaptigraph_amd.graph import threshold_sq                                 # noqa: E402
from adaptigraph_amd.model import DynamicsPredictor                            # noqa: E402
from adaptigraph_amd.plan_utils import decode_action                           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--material", default="rope")
    ap.add_argument("--particles", type=int, default=1000)
    ap.add_argument("--precision", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    mat, T, B = a.material, a.steps, a.samples
    task = configs.task_config(mat)
    mm = synth.MATERIALS[mat]
    w = dict(np.load(os.path.join(ROOT, "tests", "golden", "weights_seed0.npz")))
    model = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(dev).eval().set_option("precision", a.precision)
    kw = dict(spacing=0.1) if mat == "rope" else {}
    state_np, act_np = synth.make_mpc_inputs(mat, a.particles, B, seed=0, len_lo=T, len_hi=T + 0.4, **kw)
    state = torch.from_numpy(state_np).to(dev)
    action = torch.from_numpy(act_np).to(dev)
    n_obj, n_t = state.shape[0], mm["n_tools"]
    n_his = task["n_his"]
    decoded, repeat = decode_action(action, push_length=task["push_length"])
    attrs, p_instance, mask, tool_mask, obj_still = _constants(B, n_obj, n_t, task["max_n"], dev)
    obj_still = obj_still.expand(B, n_obj, 3)
    phys = torch.full((B, 1), 0.5, device=dev)
    thr = threshold_sq(mm["radius"], B, dev, _lib.AG_VARIANT_BATCH)
    y = state[:, 1].min().expand(B)
    eef, dlt, raise_by = _place_tool_lean(task, decoded[:, 0], action[:, 0, 2], y, dev, zero=obj_still[:, 0, 0])
    obj = state[None].expand(B, n_obj, 3)

    def run(eef_, t, mask_=None):
        state0 = torch.cat([obj[:, None].expand(B, n_his, n_obj, 3), eef_[:, None].expand(B, n_his, n_t, 3)], dim=2).contiguous()
        delta = torch.cat([obj_still, dlt], dim=1).contiguous()
        rep = torch.full((B,), t, dtype=torch.int32, device=dev)
        _, fin = rollout(model, state0, delta, attrs, p_instance, phys, mask if mask_ is None else mask_, tool_mask, thr, rep, t, mm["topk"], mm["connect_tools_all"], n_t,
                         _lib.AG_HEIGHT_MIN, None, raise_by, return_state=True)
        return fin[:, -1, :n_obj].clone()

    # the base trajectory: the same rollout with the tool slots INVALID (mask False: no edge touches them).  (Parking the tool far away is not enough:
    # forward_dynamics.py:163-168 puts the tool back at the cloud's height after every step.)
    no_tool = mask.clone()
    no_tool[:, n_obj:] = False
    lines = [f"# shared-state probe: {mat}-{a.particles}+{n_t} tool(s), {B} sampled pushes x {T} model steps, precision mode {a.precision}, seed-0 weights",
             "# clean = position bit-equal to the same step of a rollout whose tool slots are masked invalid (the base trajectory: no tool edges);",
             "# base_spread = distinct base positions over the samples (0: the parked runs of all samples agree bit for bit -> ONE base trajectory serves all)",
             "# step  clean_pairs  of  clean_frac  samples_all_clean  samples_all_dirty  max_dirty_per_sample  base_spread"]
    tot_clean = tot = 0
    for t in range(1, T + 1):
        real, base = run(eef, t), run(eef, t, no_tool)
        eq = (real.view(torch.int32) == base.view(torch.int32)).all(-1)            # (B, n_obj)
        spread = int((base.view(torch.int32) != base[:1].view(torch.int32)).any(-1).sum().item())
        dirty = (~eq).sum(1)
        c, n = int(eq.sum().item()), eq.numel()
        tot_clean += c
        tot += n
        lines.append(f"{t:5d}  {c:11d}  {n:9d}  {c / n:9.4f}  {int((dirty == 0).sum()):17d}  {int((dirty == n_obj).sum()):17d}  {int(dirty.max()):20d}  {spread:11d}")
    lines.append(f"# all steps: clean {tot_clean} of {tot} = {tot_clean / tot:.4f}   (status {model.take_status()})")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
