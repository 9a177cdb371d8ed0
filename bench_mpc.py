#!/usr/bin/env python3
"""Side benchmark for BASELINE configs[4]: "MPC planning loop: 1024 sampled action sequences x 15-step rollout on
rope, per-iteration wall-clock".  (bench.py stays the driver's headline line; this one measures the "next" row n1.)

    python bench_mpc.py --gpus 1 --steps 5 --warmup 2 [--particles 1000] [--samples 1024] [--push-steps 15]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench_mpc.py --gpus N ...

One "step" = one MPPI iteration exactly as planner.py's trajectory_optimization_mppi runs it: sample `--samples`
action sequences around the current one, roll every one of them out (`--push-steps` model steps: edge rebuild + GNN
forward + tool update each), chamfer cost to the target + push-start penalty + workspace-box penalty, softmax
update.  Strong scaling: the samples shard across ranks, one RCCL all-gather of the predicted states, every rank
evaluates the (cheap) cost and update redundantly.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
from functools import partial

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL across processes needs it)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from adaptigraph_amd import _lib, configs, losses, mpc, synth          # noqa: E402
from adaptigraph_amd.model import DynamicsPredictor                    # noqa: E402


def mppi_bench(dev, particles=1000, samples=1024, push_steps=15, steps=5, warmup=2, precision="fast", world=1, chunk=None, streams=None, dry=False,
               shared_state=0):
    """Time `steps` MPPI iterations (sample -> sharded rollout -> chamfer/penalty cost -> softmax update) on this rank's GPU.
    Returns (ms per iteration measured on this rank, last reward tensor)."""
    import time
    mat = "rope"
    task = configs.task_config(mat)
    # every sampled push is exactly push_steps long so an iteration is a fixed amount of work
    lo = np.array(task["action_lower_lim"], np.float32)
    hi = np.array(task["action_upper_lim"], np.float32)
    lo[3], hi[3] = push_steps, push_steps + 0.5
    state, act = synth.make_mpc_inputs(mat, particles, 1, seed=0, len_lo=push_steps, len_hi=push_steps + 0.4,
                                       spacing=0.1 if particles >= 500 else 0.2)
    target = (state + np.array([0.4, 0.0, 0.3], np.float32)).astype(np.float32)
    bbox = np.array([[state[:, 0].min() - 5, state[:, 0].max() + 5], [state[:, 2].min() - 5, state[:, 2].max() + 5]])
    g = torch.Generator().manual_seed(0)
    ppm = configs.ppm_optimizer_stub(mat)
    ppm.physics_param = {mat: torch.tensor([0.5], device=dev)}
    state_t, target_t = torch.from_numpy(state).to(dev), torch.from_numpy(target).to(dev)
    if dry:      # --dry-run: no engine, no GPU — sampling, sharding, the all-gather and the MPPI update on CPU tensors with a stand-in rollout and cost
        model, error = None, (lambda s: (s - target_t[None]).square().sum(-1).mean(-1))
    else:
        model = DynamicsPredictor(configs.model_config(), configs.material_config(mat), configs.dataset_config(mat), dev)
        with torch.no_grad():
            for p in model.parameters():
                p.copy_(torch.empty_like(p).uniform_(-1, 1, generator=g) / np.sqrt(p.shape[-1]))
        model = model.to(dev).eval().set_option("precision", {"f32": 0, "bf16x3": 1, "fast": 2}[precision])
        if streams is not None:
            model.set_option("rollout_streams", streams)
        model.set_option("shared_state", shared_state)      # 1: the samples share one cloud — roll the tool-less base out once, per sample only what can differ (bit-identical)
        error = partial(losses.chamfer, y=target_t[None])
    planner = mpc.MPPIPlanner(model, dev, ppm, error,
                              partial(losses.rope_penalty, sim_real_ratio=task["sim_real_ratio"]), bbox, lo, hi,
                              n_sample=samples, n_update_iter=1, rollout_best=False, n_sample_chunk=chunk, shared_state=None if dry else bool(shared_state))
    if dry:
        from adaptigraph_amd.dist import dynamics_sharded

        def stub(st, acts):
            return {"state_seqs": (acts[:, :, :3].sum(-1)[:, :, None, None] * 1e-3 + st[None, None]).float(), "action_seqs": acts * 2.0}
        planner.model_rollout = lambda st, acts, copy=True: dynamics_sharded(stub, st, acts, copy=copy)
    act_seq = torch.from_numpy(act[0]).to(dev)

    def one_iteration(seq, it):
        torch.manual_seed(1234 + it)
        smp = planner.sample(seq, 1)                      # iter_index > 0: perturb around the current sequence; broadcast from rank 0
        new_seq, reward, _ = planner.step(state_t, smp)
        return new_seq, reward

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    seq = act_seq
    for i in range(warmup):
        seq, _ = one_iteration(seq, i)
    fence()
    w0 = time.perf_counter()
    for i in range(steps):
        seq, reward = one_iteration(seq, warmup + i)
    fence()
    return (time.perf_counter() - w0) * 1e3 / steps, reward


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=1000)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--push-steps", type=int, default=15)
    ap.add_argument("--precision", default="fast", choices=["f32", "bf16x3", "fast"])
    ap.add_argument("--chunk", type=int, default=None, help="evaluate the samples in chunks of this size, as the reference planner does "
                                                          "(config/planning/rope.yaml: n_sample 20000, n_sample_chunk 500); default: one rollout")
    ap.add_argument("--streams", type=int, default=None, help="rollout_streams engine option (default: the engine's choice)")
    ap.add_argument("--shared-state", type=int, default=1, choices=[0, 1],
                    help="engine option shared_state (default 1 here: an MPPI iteration rolls ONE cloud out under all sampled pushes — the engine rolls the "
                         "tool-less base trajectory out once and computes per sample only the rows that can differ; same bits as 0)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU, no engine: gloo process group on CPU tensors, sampling / sharding / all-gather / MPPI "
                                                             "update and the JSON line only (contract check; the number is meaningless)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.dry_run:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench_mpc.py needs an MI355X: the engine has no CPU path")
        local %= max(1, torch.cuda.device_count())       # fewer GPUs than ranks (2-rank test on one GPU): wrap around
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("AG_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; tests on a 1-GPU box use "gloo"
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
        _lib.lib()
    per_it, _ = mppi_bench(dev, a.particles, a.samples, a.push_steps, a.steps, a.warmup, a.precision, world, a.chunk, a.streams, dry=a.dry_run,
                           shared_state=a.shared_state)
    tt = torch.tensor([per_it], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ms = float(tt.item()) * a.steps
    if rank == 0:
        per_it = ms / a.steps
        print(json.dumps({
            "metric": "MPPI iteration wall-clock (sample + rollout + cost + update)", "value": round(per_it, 3), "unit": "ms",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(per_it, 3),
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": a.precision,
            "data": "dry-run (no engine: contract check only)" if a.dry_run else "synthetic",
            "graph_steps_per_s": round(a.samples * a.push_steps / per_it * 1e3, 1),
            "config": {"workload": f"rope-{a.particles}+1 MPPI: {a.samples} samples x {a.push_steps}-step rollout, chamfer "
                                   f"cost to a {a.particles}-point target", "samples": a.samples, "push_steps": a.push_steps,
                       "particles": a.particles, "parallelism": f"samples/{world}", "sample_chunk": a.chunk, "rollout_streams": a.streams,
                       "shared_state": a.shared_state}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
